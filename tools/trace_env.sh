#!/bin/bash
# kernel trace of tools/ab_env.py (envelopes in motion, note traffic): per-kernel durations in those regimes
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_env -- python3 $R/tools/ab_env.py > $R/gpurun_out/tr_env.log 2>&1
cat $R/gpurun_out/tr_env/*/*kernel_stats.csv | head -8
