#!/bin/bash
# tools/round4.sh profiles -- the rocprofv3 passes behind profiles/r04_* (run on the GPU box through gpurun; fold here afterwards
# with `python3 tools/fold_profiles.py r04 r04_v1`).  Every pass is bounded and the steps are joined with &&.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
case "${1:-profiles}" in
  profiles)
    for w in c1 c2 c3 c4; do bash $R/tools/profile_round.sh $w r04_$w; done &&
    bash $R/tools/profile_round.sh c3 r04_c3f64 --frames 64 &&
    bash $R/tools/profile_round.sh c3 r04_c3f4800 --frames 4800 &&
    bash $R/tools/profile_round.sh c2 r04_shard17 --voices 131072 &&
    bash $R/tools/profile_round.sh c2 r04_shard18 --voices 262144 &&
    bash $R/tools/profile_round.sh c2 r04_shard19 --voices 524288 ;;
  mid)
    bash $R/tools/profile_mid.sh r04m_shard17 &&
    bash $R/tools/profile_mid.sh r04m_c2 --workload c2 --voices 65536 &&
    bash $R/tools/profile_mid.sh r04m_c1 --workload c1 --voices 4096 ;;
esac
