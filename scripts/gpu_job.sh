O=gpurun_out/r3f; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_user_targets.py -m gpu -q 2>&1 | tail -60 ) > $O/pytest_user.log
tail -5 $O/pytest_user.log
bash scripts/profile_head.sh cfg2 2>&1 | tail -3
