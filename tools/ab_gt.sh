# A/B of library builds (_ab/libskred_<name>.so, e.g. make OBJ=/tmp/o OUT=$PWD/_ab/libskred_x.so EXTRA=-DSK_GT_RANKS=16) of the
# in-place steady kernel under note traffic (tools/ab_gain.py): per-kernel durations from a rocprofv3 trace.
# usage: bash tools/ab_gt.sh <name> ...   (the tree's own build is always measured first)
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in base "$@"; do
  if [ $v = base ]; then unset SKRED_AMD_LIB; else export SKRED_AMD_LIB=$R/_ab/libskred_$v.so; fi
  for k in 104 524; do
    rm -rf $R/gpurun_out/abgt_$v
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abgt_$v -- python3 $R/tools/ab_gain.py $k > $R/gpurun_out/abgt_$v.log 2>&1 || exit 1
    echo "== $v  $k voices per block: $(tail -1 $R/gpurun_out/abgt_$v.log)"
    python3 -c "
import csv,glob,sys
for r in csv.DictReader(open(glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0])):
    if 'sk_gain' in r['Name'] or 'sk_render_fast2' in r['Name']: print('   %-70s calls %4s avg %8.1f us  min %8.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
" $R/gpurun_out/abgt_$v
    rm -rf $R/gpurun_out/abgt_$v
  done
done
