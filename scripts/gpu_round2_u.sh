timeout 600 python scripts/hier_bench.py 2>&1 | tail -3
AHMC_NUTS_BATCH=8 timeout 600 python scripts/hier_bench.py 2>&1 | tail -2
