#!/bin/bash
# rocprofv3 passes over the bench command at HEAD (run on the GPU box through gpurun):
#   gpurun --timeout 1500 -- 'bash scripts/profile_head.sh cfg2 [cfg3 ...]'
# kernel trace + stats in one pass; PMC counters in their own passes with --kernel-trace only (never combined with
# sys/hip/hsa traces).  scripts/profile_counters.py then distils profiles/counters_at_head.json (what bench.py reads,
# keyed on the digest of the kernel sources) and the per-config kernel tables.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for CFG in "$@"; do
  O=$R/gpurun_out/prof_$CFG
  rm -rf $O; mkdir -p $O
  STEPS=${PROFILE_STEPS:-20}
  # one timed run, no throw-away engine, no ESS leg: every launch of the dominant kernels belongs to the reported run
  # PROFILE_EXTRA: more bench arguments (e.g. "--transitions-per-step 4" for a run short enough for PMC passes over the dense
  # engine's hundreds of dispatches per transition)
  CMD="python bench.py --config $CFG --steps $STEPS --warmup 0 --repeats 1 --ess 0 --no-cpu-baseline ${PROFILE_EXTRA:-}"
  echo "$CMD" > $O/cmd.txt
  timeout 600 $CMD > $O/bench_plain.json 2> $O/bench_plain.err
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/kt.json 2> $O/kt.err
  # PROFILE_PASSES bounds the run for the dense configs (hundreds of thousands of dispatches per pass: the five
  # databases of a 4-step cfg4 run filled the box's disk)
  P=" ${PROFILE_PASSES:-fetch write sq1 sq2 sq3 grbm} "
  pmc() { name=$1; shift; case "$P" in *" $name "*) ;; *) return;; esac
          timeout ${PROFILE_PASS_TIMEOUT:-900} rocprofv3 --pmc "$@" --kernel-trace -d $O/$name -o $name -- $CMD > $O/$name.json 2> $O/$name.err
          find $O/$name -name "*.csv" -size +8M -delete; }
  pmc fetch FETCH_SIZE
  pmc write WRITE_SIZE
  pmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_F64
  pmc sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
  # the dynamic instruction MIX of the VALU stream (what the mix-weighted issue roof is made from, scripts/probe/valu_rate.hip)
  pmc sq3 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
  pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
  python scripts/profile_counters.py $O $CFG > $O/summary.json 2> $O/summary.err
  # keep the merge-back small: the per-dispatch databases stay on the box
  find $O -name "*.db" -size +8M -delete
  find $O -name "*_kernel_trace.csv" -size +8M -delete
  du -sh $O > $O/size.txt
  tail -c 2500 $O/summary.json
done
