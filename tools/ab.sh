# same-box A/B of bench.py under environment switches: tools/ab.sh "A=0" "HDMOE_X=0" ...   (each argument = one run's env assignments)
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-sampler 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stage_ms']
print('%-44s %.3f ms  pre %.2f unet %.2f vit %.2f post %.2f unet_bwd %.2f vit_bwd %.2f ur_bwd %.2f pre_bwd %.2f' % (sys.argv[1], d['ms_per_step'], s['pre'][1]-s['pre'][0], s['unet'][1]-s['unet'][0], s['vit'][1]-s['vit'][0], s['post'][1]-s['post'][0], s['unet_bwd'][1]-s['unet_bwd'][0], s['vit_bwd'][1]-s['vit_bwd'][0], s['ur_bwd'][1]-s['ur_bwd'][0], s['pre_bwd'][1]-s['pre_bwd'][0]))" "$1"; }
for a in "$@"; do run "$a"; done
