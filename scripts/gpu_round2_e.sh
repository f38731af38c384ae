O=gpurun_out/r2e; mkdir -p $O
LIBC_FATAL_STDERR_=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "multiwave" > $O/mw_alone.log 2>&1; echo "exit $?" >> $O/mw_alone.log; grep -v "^  File\|^Thread" $O/mw_alone.log | tail -8 | cut -c1-300
LIBC_FATAL_STDERR_=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/parity_file.log 2>&1; echo "exit $?" >> $O/parity_file.log; grep -v "^  File\|^Thread" $O/parity_file.log | tail -12 | cut -c1-300
timeout 900 python -m pytest tests -q -m gpu -rf --timeout 600 --deselect "tests/test_gpu_parity.py::test_multiwave_chains" > $O/gpu_suite.log 2>&1; echo "exit $?" >> $O/gpu_suite.log; tail -8 $O/gpu_suite.log | cut -c1-300
timeout 600 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -c 2500 $O/bench_cfg2.json
