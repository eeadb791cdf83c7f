set -e
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
for i in 1 2; do
HDMOE_TRUNK_FUSED=0 timeout -k 10 300 python bench.py --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('layers', d['ms_per_step'], d['config'].get('stage_ms'))"
HDMOE_TRUNK_FUSED=1 timeout -k 10 300 python bench.py --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused ', d['ms_per_step'], d['config'].get('stage_ms'))"
done
