O=$PWD/gpurun_out/r2m; mkdir -p $O
timeout 900 python scripts/stress_multiwave.py > $O/stress.log 2>&1; echo "exit $?" >> $O/stress.log; grep -v "^ok " $O/stress.log | tail -15; grep -c "^ok " $O/stress.log
