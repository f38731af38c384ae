(cd _old_r1 && timeout 600 python scripts/hier_bench.py 2>&1 | tail -2)
(cd _old_r1 && AHMC_DEBUG=1 STEPS=8 ADAPT=20 timeout 600 python scripts/hier_bench.py 2>&1 | grep "k_nuts<" | sort | uniq -c | head -5)
AHMC_DEBUG=1 STEPS=8 ADAPT=20 timeout 600 python scripts/hier_bench.py 2>&1 | grep "k_nuts<" | sort | uniq -c | head -5
