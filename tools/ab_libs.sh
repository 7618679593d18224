#!/bin/bash
# tools/ab_libs.sh SCRIPT VARIANT...  -- run one measuring script once per library build under _ab/<variant>/ (same box,
# same session), twice in alternation so that drift shows.  Output: gpurun_out/ab_libs.txt
R=${GRAFT_REPO_ROOT:-$PWD}; S=$1; shift  # (e.g. "ab.py steady --sizes c2:65536": quote a script with arguments)
mkdir -p $R/gpurun_out; : > $R/gpurun_out/ab_libs.txt
for round in 1 2; do
  for v in "$@"; do
    SKRED_AMD_LIB=$R/_ab/$v/libskred_amd.so timeout -k 10 150 python3 $R/tools/$S >> $R/gpurun_out/ab_libs.txt 2>&1 || exit 1
  done
done
cat $R/gpurun_out/ab_libs.txt
