run() { env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stage_ms']; print('$*', d['ms_per_step'], d['config']['loss'], {k: round(v[1]-v[0],2) for k,v in s.items()})"; }
run HDMOE_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4
run HDMOE_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=8
run HDMOE_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4
run HDMOE_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=8
run HDMOE_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=6
run X=1
