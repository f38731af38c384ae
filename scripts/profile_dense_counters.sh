#!/bin/bash
# HBM bytes and instruction counters of the dense engine's kernels (k_dgemm, k_d_tree, k_dgemm_small) on a SHORT cfg4-shaped job
# (scripts/dense_split_check.py: 24 adapting transitions + 2 x 1 draws at 8 192 x 512, ≈ 40 000 dispatches) — the bench-sized
# run has ≈ 120 000 per pass and did not finish its PMC passes inside a GPU call (round 2).
#   gpurun --timeout 600 -- 'bash scripts/profile_dense_counters.sh'   →  gpurun_out/dense_pmc/summary.json
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/dense_pmc; rm -rf $O; mkdir -p $O
CMD="python scripts/dense_split_check.py"
export SKIP_SMALL=1 MODES=${MODES:-1} ADAPT=${ADAPT:-24} STEPS=${STEPS:-1}
timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/kt.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o fetch -- $CMD > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o write -- $CMD > $O/write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_F64 --kernel-trace -d $O/sq -o sq -- $CMD > $O/sq.log 2>&1
python scripts/pmc_by_kernel.py $O/kt $O/fetch $O/write $O/sq > $O/summary.json 2> $O/summary.err
python scripts/dense_timeline.py $O/kt > $O/timeline.json 2>> $O/summary.err
find $O -name "*.db" -size +4M -delete; find $O -name "*.csv" -size +4M -delete
grep "cfg4 shard" $O/*.log | cut -c1-200; head -c 1500 $O/summary.json
