run() { env "$@" timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stage_ms']; print('$*', d['ms_per_step'], {k: round(v[1]-v[0],2) for k,v in s.items() if k in ('ur','vit','unet','unet_bwd','vit_bwd')})"; }
for i in 1 2 3; do
run HDMOE_C6_G=256
run HDMOE_C6_G=224
run HDMOE_C6_G=192
done
