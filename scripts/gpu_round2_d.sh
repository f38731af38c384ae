O=gpurun_out/r2d; mkdir -p $O
bash scripts/ab_bench.sh $O base exp2 ds1 ds3 exp2ds3 base@b64:AHMC_NUTS_BATCH=64 base@b128:AHMC_NUTS_BATCH=128 2>&1 | tee $O/ab.log
PYTHONFAULTHANDLER=1 AMD_LOG_LEVEL=1 timeout 900 python -m pytest tests -q -m gpu -rf --timeout 600 -v > $O/gpu_suite.log 2>&1; echo "exit $?" >> $O/gpu_suite.log; grep -v PASSED $O/gpu_suite.log | tail -40 | cut -c1-300
