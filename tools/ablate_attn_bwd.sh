# Builds libamk variants with parts of the fused f32 attention backward's tile loop switched off (AMK_BWD_ABL bits, see
# csrc/attn_bwd_fused.hip) and times each at the ViT-VQGAN layer shape.  Results are WRONG by construction; timing only.
#   bash tools/ablate_attn_bwd.sh   (on the GPU box, through gpurun)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/attention-models_amd/csrc
mkdir -p build/abl
OTHERS=$(ls build/*.o | grep -v "attn_bwd_fused.o")
for n in ${VARIANTS:-0 1 2 3 4 8 16 32 64 7 15 63 127}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DAMK_BWD_ABL=$n -c attn_bwd_fused.hip -o build/abl/attn_bwd_fused_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libamk_b$n.so build/abl/attn_bwd_fused_$n.o $OTHERS
  echo "== AMK_BWD_ABL=$n"
  AMK_LIB=$PWD/build/abl/libamk_b$n.so timeout -k 10 120 python3 $R/tools/kbench_attn_bwd.py 2>&1 | grep -E "keys=256 kept  |keys=256 recompute  "
done
