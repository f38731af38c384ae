python scripts/ext_trace.py 16384; python scripts/ext_trace.py 1024
