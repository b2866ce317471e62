# PMC passes (SQ + GRBM set, LDS set, FETCH_SIZE, WRITE_SIZE: one rocprofv3 run each, never combined with other trace
# domains) over a micro-benchmark, folded by tools/pmc_quick.py into OUT/digest.txt.  Run on the GPU box through gpurun:
#   OUT=r3_pmc_dense bash tools/pmc_dense.sh                       (dense GEMMs: tools/kbench_dense.py)
#   OUT=r3_pmc_bf16 SCRIPT=tools/kbench_attn_bf16.py ARGS="--iters 5" FILTER="attn_bf16 attn_fwd_kernel attn_bwd_fused" bash tools/pmc_dense.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${OUT:-r3_pmc_dense}
SCRIPT=${SCRIPT:-tools/kbench_dense.py}
ARGS=${ARGS:---only kv,ffn_w12,ffn_w3 --no-fused --no-check --iters 3}
FILTER=${FILTER:-gemm_ Cijk}
cd /tmp && export TMPDIR=/tmp
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmcA -o a -- python3 $R/$SCRIPT $ARGS > $O/pmcA.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmcE -o e -- python3 $R/$SCRIPT $ARGS > $O/pmcE.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcB -o b -- python3 $R/$SCRIPT $ARGS > $O/pmcB.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcC -o c -- python3 $R/$SCRIPT $ARGS > $O/pmcC.log 2>&1
for f in a e b c; do
  C=$(find $O -name "${f}_counter_collection.csv" | head -1)
  echo "== pass $f"; python3 $R/tools/pmc_quick.py $C $FILTER
done > $O/digest.txt
find $O -name "*.csv" -delete
