# the bulk phase of the dense engine under the kernel trace (after a real step-size adaptation), AHMC_DENSE_SPLIT=1 and 0
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2y; mkdir -p $O
for m in 1 0; do
  SKIP_SMALL=1 MODES=$m ADAPT=24 STEPS=3 timeout 40 rocprofv3 --kernel-trace -d $O/tl$m -o tl -- python scripts/dense_split_check.py > $O/run$m.log 2>&1
  python scripts/dense_timeline.py $O/tl$m > $O/timeline_split$m.json 2> $O/timeline_split$m.err
  rm -rf $O/tl$m
done
