run() { echo "== $*"; env "$@" HDMOE_BENCH_TRACE_LOSS=1 timeout -k 10 200 python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline $EXTRA 2>gpurun_out/probe.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['loss'], d['config']['launch'][:30])"; grep " loss " gpurun_out/probe.err | sed 's/\[bench\]//' | tr '\n' ' '; echo; }
EXTRA="" run HDMOE_BENCH_FORCE_DIST=1 HDMOE_BENCH_NO_FORCE_COLL=1 HDMOE_BENCH_SKIP_BARRIER=1
EXTRA="" run HDMOE_BENCH_FORCE_DIST=1 HDMOE_BENCH_NO_FORCE_COLL=1 HDMOE_BENCH_EARLY_BARRIER=1
EXTRA="" run HDMOE_BENCH_FORCE_DIST=1 HDMOE_BENCH_EARLY_BARRIER=1
