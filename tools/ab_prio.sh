run() { env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stage_ms']; print('$*', d['ms_per_step'], d['config']['loss'], {k: round(v[1]-v[0],2) for k,v in s.items()})"; }
run HDMOE_W6_PARTS=256
run HDMOE_W6_PARTS=128
run HDMOE_W6_PARTS=96
run HDMOE_W6_PARTS=160
run HDMOE_W6_PARTS=128 HDMOE_W6_PARTS_SPLIT=512
run HDMOE_W6_PARTS=128 HDMOE_W6_PARTS_SPLIT=128
