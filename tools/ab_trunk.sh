set -e
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fused_router_trunk or router_golden or staged_step or split" > gpurun_out/trunk.log 2>&1 || { tail -40 gpurun_out/trunk.log; exit 1; }
tail -2 gpurun_out/trunk.log
for i in 1 2; do
for f in 0 1; do
HDMOE_TRUNK_FUSED=$f timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stage_ms']; print('fused=$f', d['ms_per_step'], 'ur', s['ur'], 'vit', s['vit'], 'ur_bwd', s['ur_bwd'], 'vit_bwd', s['vit_bwd'], 'unet_bwd', s['unet_bwd'])"
done
done
