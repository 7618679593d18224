#!/bin/bash
# tools/profile_round.sh WORKLOAD TAG [bench args...] -- the rocprofv3 passes behind profiles/ (run on the GPU box through gpurun).
# Extra bench args select another shape of the same workload (e.g. `--frames 64`: the low-latency block of config 3).
# Writes gpurun_out/prof_TAG/{trace,fetch,write,sq,mem}/ ; every pass is bounded (timeout -k) and joined with &&.
set -e
W=${1:-c3}; TAG=${2:-r01}; shift; shift || true; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --workload $W --no-cpu --no-extra --time-every 1 $@"   # every launch bracketed; one dispatch = one block (render + mix-down + master volume)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 100 --warmup 20 > $O/trace.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 5 --warmup 5 > $O/fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 5 --warmup 5 > $O/write.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq -- $B --steps 5 --warmup 5 > $O/sq.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $O/mem -- $B --steps 5 --warmup 5 > $O/mem.log 2>&1
echo "profile passes done: $O"
