#!/bin/bash
# tools/profile_mid.sh TAG [bench args...] -- SQ counter passes for one bench configuration (default: the strong-scaling
# shard, 131072 voices of the C2 recipe on the one-voice kernel).  Run on the GPU box through gpurun.
set -e
TAG=${1:-mid}; shift || true
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prof_$TAG
ARGS=${@:---workload c2 --voices 131072 --fast2-min-voices 2000000000}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py $ARGS --no-cpu --no-low-latency --no-fixed-point --time-every 1"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 50 --warmup 10 > $O/trace.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq -- $B --steps 5 --warmup 5 > $O/sq.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/sq2 -- $B --steps 5 --warmup 5 > $O/sq2.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/sq3 -- $B --steps 5 --warmup 5 > $O/sq3.log 2>&1
echo "profile passes done: $O"
