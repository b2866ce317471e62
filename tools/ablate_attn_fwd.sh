# Builds libamk variants with parts of the f32 attention forward's tile loop switched off (AMK_FWD_ABL bits, see
# csrc/attn_fwd.hip) and times each at the ViT-VQGAN layer shape: what each part of the loop costs beside the MFMAs.
# Results are WRONG by construction; timing only.   bash tools/ablate_attn_fwd.sh   (on the GPU box, through gpurun)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/attention-models_amd/csrc
mkdir -p build/abl
OTHERS=$(ls build/*.o | grep -v "attn_fwd.o")
for n in ${VARIANTS:-0 1 2 4 8 16 32 64 128 7 15 48 112 255}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DAMK_FWD_ABL=$n -c attn_fwd.hip -o build/abl/attn_fwd_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libamk_f$n.so build/abl/attn_fwd_$n.o $OTHERS
  echo "== AMK_FWD_ABL=$n"
  AMK_LIB=$PWD/build/abl/libamk_f$n.so timeout -k 10 120 python3 $R/tools/kbench_attn_fwd.py --f32-only 2>&1 | tail -1
done
