run() { D=$1 N=$2 AHMC_GEOMETRY=$3 ADAPT=60 STEPS=16 timeout 300 python scripts/hier_bench.py 2>&1 | grep "cfg5" | sed "s/^/[$3] /"; }
run 256 262144 64,4
run 256 262144 128,2
run 512 131072 64,8
run 512 131072 128,4
run 512 131072 256,2
run 1024 65536 128,8
run 1024 65536 256,4
run 2048 32768 256,8
run 2048 32768 512,4
