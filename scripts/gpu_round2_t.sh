O=$PWD/gpurun_out/r2t; mkdir -p $O
bash scripts/ab_bench.sh $O base fuse 2>&1 | tee $O/ab.log
