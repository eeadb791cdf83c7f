set -e
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "trunk or router_golden or staged or batch_independence or real_widths" > gpurun_out/t.log 2>&1 || { tail -40 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
run() { env "$@" timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stage_ms']; print('$*', d['ms_per_step'], d['config']['loss'], {k: round(v[1]-v[0],2) for k,v in s.items() if k in ('ur','vit','unet','unet_bwd','vit_bwd')})"; }
for i in 1 2 3; do
run HDMOE_TRUNK_FIN_IN_CONV=0
run HDMOE_TRUNK_FIN_IN_CONV=1
done
