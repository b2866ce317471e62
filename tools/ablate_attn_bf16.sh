# Builds libamk variants with parts of the bf16 attention backward switched off (A16_ABLATE bits: 1 no dQ stores,
# 2 no dQ product, 4 no dV / dK products, 8 no exp) and times each: where the tile time goes.  Results are WRONG by
# construction; timing only.   bash tools/ablate_attn_bf16.sh   (on the GPU box, through gpurun)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/attention-models_amd/csrc
mkdir -p build/abl
OTHERS=$(ls build/*.o | grep -v attn_bf16.o)
for n in ${VARIANTS:-1 2 4 8 6 14}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DA16_ABLATE=$n -c attn_bf16.hip -o build/abl/attn_bf16_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libamk_$n.so build/abl/attn_bf16_$n.o $OTHERS
  echo "== A16_ABLATE=$n"
  AMK_LIB=$PWD/build/abl/libamk_$n.so timeout -k 10 120 python3 $R/tools/kbench_attn_bf16.py --no-f32 2>&1 | tail -1
done
