export SR_BENCH_PROJ=1 SR_BENCH_GATHER=1
for dbg in 0 2048 256 2304 0; do echo -n "dbg=$dbg " ; SR_CONV_DBG=$dbg timeout -k 10 100 python tools/conv_bench.py --only up2 || exit 1; done
