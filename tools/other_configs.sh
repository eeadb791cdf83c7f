for c in 1 3 4; do
timeout -k 10 400 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-sampler 2>gpurun_out/cfg$c.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', d['ms_per_step'], d['config']['loss'], d['config']['launch'][:40])" || tail -5 gpurun_out/cfg$c.err
done
timeout -k 10 400 python tools/sampler_bench.py 2>&1 | tail -4
