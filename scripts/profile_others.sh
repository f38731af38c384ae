#!/bin/bash
# rocprofv3 kernel-trace summaries of the cfg3 / cfg4 / cfg5 scripts (top kernels) -> gpurun_out/prof_others/*.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_others
rm -rf $O; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" > /dev/null 2>&1; }
prof() {
  name=$1; shift
  ( env "$@" rocprofv3 --kernel-trace --stats -d $O/$name -o $name -- python $SCRIPT > $O/$name.log 2>&1 )
  python scripts/top_kernels.py $O/$name > $O/$name.top.txt 2>&1
  grep -h "cfg" $O/$name.log | tail -2 >> $O/$name.top.txt
  rm -rf $O/$name
}
SCRIPT=scripts/funnel_bench.py prof cfg3 X=1
SCRIPT=scripts/dense_bench.py prof cfg4 ADAPT=30 STEPS=32
SCRIPT=scripts/hier_bench.py prof cfg5 ADAPT=60 STEPS=32
cat $O/*.top.txt
