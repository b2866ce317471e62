# Builds libamk variants with parts of gemm_bf16_kernel switched off (G16_ABLATE bits: 1 no epilogue, 2 no MFMAs,
# 4 no operand reads from LDS) and times the forward shapes: where a tile's time goes.  Results are WRONG by
# construction; timing only.   bash tools/ablate_gemm_bf16.sh   (on the GPU box, through gpurun)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/attention-models_amd/csrc
mkdir -p build/abl
OTHERS=$(ls build/*.o | grep -v gemm_bf16.o)
for n in ${VARIANTS:-0 1 2 3 4 5 6 7}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DG16_ABLATE=$n -c gemm_bf16.hip -o build/abl/gemm_bf16_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libamk_g16_$n.so build/abl/gemm_bf16_$n.o $OTHERS
  echo "== G16_ABLATE=$n"
  AMK_LIB=$PWD/build/abl/libamk_g16_$n.so timeout -k 10 120 python3 $R/tools/kbench_tn_bf16.py 2>&1 | grep -E "NT w12|NN w12|NT kv|gate" | cut -c1-60
done
