# rocprofv3 evidence of a round (run on the GPU box through gpurun; ROUND=r02 by default):
#   kernel-trace stats of the bench command + four separate --pmc passes over the kernel micro-benchmarks
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r04}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/${ROUND}prof
mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-variants > $O/bench_profiled.json 2> $O/bench_profiled.err
echo "stats pass done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmcA -o a -- python3 $R/tools/kbench.py --batch 32 --iters 5 > $O/pmcA.log 2>&1
echo "pmc A done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcB -o b -- python3 $R/tools/kbench.py --batch 32 --iters 5 > $O/pmcB.log 2>&1
echo "pmc B done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcC -o c -- python3 $R/tools/kbench.py --batch 32 --iters 5 > $O/pmcC.log 2>&1
echo "pmc C done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum --output-format csv -d $O/pmcD -o d -- python3 $R/tools/kbench.py --batch 32 --iters 5 > $O/pmcD.log 2>&1
echo "pmc D done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/pmcE -o e -- python3 $R/tools/kbench.py --batch 32 --iters 5 > $O/pmcE.log 2>&1 || echo "pmc E failed (optional)"
rm -f $O/*/*_kernel_trace.csv.bak
find $O -name "*.csv" | head -40
