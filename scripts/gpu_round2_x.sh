# kernel-trace timelines of the dense engine's schedules (AHMC_DENSE_SPLIT=1: a stream per chain half; 2: a stream per kernel kind)
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2x; mkdir -p $O
for m in 1 2; do
  SKIP_SMALL=1 MODES=$m ADAPT=3 STEPS=2 timeout 45 rocprofv3 --kernel-trace -d $O/tl$m -o tl -- python scripts/dense_split_check.py > $O/run$m.log 2>&1
  python scripts/dense_timeline.py $O/tl$m > $O/timeline_split$m.json 2> $O/timeline_split$m.err
  rm -rf $O/tl$m
done
tail -3 $O/run1.log $O/run2.log | cut -c1-200
