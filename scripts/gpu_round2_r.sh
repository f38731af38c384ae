timeout 200 python scripts/dbg_forcedist.py; echo "exit $?"
