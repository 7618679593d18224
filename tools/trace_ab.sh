#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
for v in envw pv2; do
  export SKRED_AMD_LIB=$R/_ab/$v/libskred_amd.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_$v -- python3 $R/bench.py --frames 64 --no-extra --no-cpu --steps 400 --warmup 100 > $R/gpurun_out/tr_$v.log 2>&1 || exit 1
done
for v in envw pv2; do echo $v; cat $R/gpurun_out/tr_$v/*/*kernel_stats.csv | head -4; done
