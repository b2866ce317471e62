set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r01final
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmcA -o a -- python3 $R/tools/kbench.py --batch 32 --iters 5 > $O/pmcA.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcB -o b -- python3 $R/tools/kbench.py --batch 32 --iters 5 > $O/pmcB.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcC -o c -- python3 $R/tools/kbench.py --batch 32 --iters 5 > $O/pmcC.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum --output-format csv -d $O/pmcD -o d -- python3 $R/tools/kbench.py --batch 32 --iters 5 > $O/pmcD.log 2>&1
rm -f $O/*/*_kernel_trace.csv.bak
ls $O/*
