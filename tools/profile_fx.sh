#!/bin/bash
# tools/profile_fx.sh TAG [ab.py fx args...] -- kernel trace + SQ counter passes of the fixed-point render kernel (default: 2^20 voices,
# biquad, linear lookup: bench.py's `fixed_point` leg).  Run on the GPU box through gpurun; fold with tools/pmc_sq.py DIR sk_fx_render.
set -e
TAG=${1:-fx}; shift || true
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prof_$TAG
ARGS=${@:---fx-sizes 1048576 --fx-filter 1 --fx-interp 1}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python3 $R/tools/ab.py fx $ARGS"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/sq3 -- $B > $O/sq3.log 2>&1
python3 $R/tools/pmc_sq.py $O sk_fx_render 30 > $O/summary.txt 2>&1 || true
cat $O/summary.txt
