#!/bin/bash
# tools/profile_patch.sh TAG PATCH SKEW -- rocprofv3 passes of one shipped-patch bank (tools/ab.py patch: banks.bank_patch tiled over
# 2^20 voices, 512-frame blocks) with the skewed blocks on (1) or off (0): kernel trace + stats, then the SQ counters in a pass of
# their own.  Run on the GPU box through gpurun; writes gpurun_out/prof_TAG/{trace,sq}/ ; bounded, joined with &&.
set -e
TAG=${1:-r04p_3sk}; P=${2:-3sk}; K=${3:-1}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python3 $R/tools/ab.py patch --patches $P --fm-skew $K"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1
echo "profile passes done: $O"
