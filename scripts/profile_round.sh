#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box through gpurun); summaries are then
# distilled into profiles/ by scripts/profile_summary.py.  Counters are collected in their own
# passes with --kernel-trace only (never combined with sys/hip/hsa traces).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd $R
CMD="python bench.py --no-cpu-baseline"
$CMD > $O/bench_plain.json 2> $O/bench_plain.err
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/bench_under_rocprof.json 2> $O/kt.err
pmc() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d $O/$name -o $name -- $CMD > $O/$name.json 2> $O/$name.err; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT
pmc sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pmc sq3 SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_MFMA_F64 SQ_IFETCH
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
find $O -name "*.csv" | head -50 > $O/files.txt
python scripts/profile_summary.py $O > $O/summary.json 2> $O/summary.err
# keep the merge-back small: drop per-dispatch traces except the ones summarised
du -sh $O > $O/size.txt
find $O -name "*_kernel_trace.csv" -size +20M -delete
tail -c 1500 $O/summary.json
