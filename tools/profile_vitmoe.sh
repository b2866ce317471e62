# rocprofv3 evidence for BASELINE.json configs[3] (ViTMoE kernels): kernel-trace stats + PMC passes over tools/kbench_moe.py
set -e
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r02}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/${ROUND}moe
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o moe -- python3 $R/tools/kbench_moe.py --iters 5 > $O/stats.log 2>&1
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmcA -o a -- python3 $R/tools/kbench_moe.py --iters 3 > $O/pmcA.log 2>&1
echo "pmc A done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcB -o b -- python3 $R/tools/kbench_moe.py --iters 3 > $O/pmcB.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcC -o c -- python3 $R/tools/kbench_moe.py --iters 3 > $O/pmcC.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum --output-format csv -d $O/pmcD -o d -- python3 $R/tools/kbench_moe.py --iters 3 > $O/pmcD.log 2>&1
echo "pmc B-D done"
rm -f $O/*/*_kernel_trace.csv.bak
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/pmcE -o e -- python3 $R/tools/kbench_moe.py --iters 3 > $O/pmcE.log 2>&1 || echo "pmc E failed (optional)"
