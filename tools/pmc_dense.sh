# PMC passes over the dense GEMM micro-benchmark (run on the GPU box through gpurun).  OUT=gpurun_out/<dir>
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${OUT:-r3_pmc_dense}
ARGS=${ARGS:---only kv,ffn_w12,ffn_w3 --no-fused --no-check --iters 3}
cd /tmp && export TMPDIR=/tmp
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmcA -o a -- python3 $R/tools/kbench_dense.py $ARGS > $O/pmcA.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmcE -o e -- python3 $R/tools/kbench_dense.py $ARGS > $O/pmcE.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcB -o b -- python3 $R/tools/kbench_dense.py $ARGS > $O/pmcB.log 2>&1
for f in a e b; do
  C=$(find $O -name "${f}_counter_collection.csv" | head -1)
  echo "== pass $f"; python3 $R/tools/pmc_quick.py $C gemm_ Cijk
done > $O/digest.txt
find $O -name "*.csv" -delete
