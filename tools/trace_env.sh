#!/bin/bash
# kernel trace of `tools/ab.py live` (envelopes in motion, note traffic): per-kernel durations in those regimes, and where the
# envelope kernel sits in time relative to the steady kernel (tools/timeline_env.py)
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/tr_env
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_env -- python3 $R/tools/ab.py live > $R/gpurun_out/tr_env.log 2>&1
cat $R/gpurun_out/tr_env/*/*kernel_stats.csv | head -8
python3 $R/tools/timeline_env.py $R/gpurun_out/tr_env 16 > $R/gpurun_out/tr_env_timeline.txt 2>&1
find $R/gpurun_out/tr_env -name "*kernel_trace.csv" -size +20M -delete
cat $R/gpurun_out/tr_env_timeline.txt
