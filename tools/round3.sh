#!/bin/bash
# tools/round3.sh -- the measurements behind profiles/r03_*: run on the GPU box through gpurun, in two calls
#   bash tools/round3.sh profiles     rocprofv3 passes for c1..c4 (tools/profile_round.sh), folded afterwards by tools/fold_profiles.py r03 r03_v1
#   bash tools/round3.sh banks        tools/measure_banks.py (all scenarios) -> gpurun_out/r03_measure_banks.txt; ta_rate / issue_rate
R=${GRAFT_REPO_ROOT:-$PWD}
case "$1" in
  profiles) for w in c1 c2 c3 c4; do bash $R/tools/profile_round.sh $w r03_$w || exit 1; done ;;
  banks) cd $R; timeout -k 10 900 python3 tools/measure_banks.py > gpurun_out/r03_measure_banks.txt 2>&1; tail -3 gpurun_out/r03_measure_banks.txt
         ./tools/ta_rate > gpurun_out/ta_rate_r03.txt 2>&1; ./tools/issue_rate > gpurun_out/issue_rate_r03.txt 2>&1 ;;
esac
