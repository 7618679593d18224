#!/bin/bash
# kernel trace + SQ counters of the fixed-point leg (bench.py default legs, the fixed-point kernels picked out)
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_fx -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu > $R/gpurun_out/tr_fx.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVES --output-format csv -d $R/gpurun_out/pm_fx -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu > $R/gpurun_out/pm_fx.log 2>&1
grep "sk_fx" $R/gpurun_out/tr_fx/*/*kernel_stats.csv
