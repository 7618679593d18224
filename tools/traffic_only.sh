#!/bin/bash
# tools/traffic_only.sh WORKLOAD TAG -- just the two HBM-traffic passes (FETCH_SIZE, WRITE_SIZE) of profile_round.sh
set -e
W=${1:-c2}; TAG=${2:-r01}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --workload $W --no-cpu --no-extra --time-every 1"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 5 --warmup 5 > $O/fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 5 --warmup 5 > $O/write.log 2>&1
echo "traffic passes done: $O"
