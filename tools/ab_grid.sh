run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['config']['stage_ms']
print('%-44s %.3f ms  unet %.2f vit %.2f post %.2f unet_bwd %.2f vit_bwd %.2f ur_bwd %.2f' % (' '.join(sys.argv[1:]), d['ms_per_step'], s['unet'][1]-s['unet'][0], s['vit'][1]-s['vit'][0], s['post'][1]-s['post'][0], s['unet_bwd'][1]-s['unet_bwd'][0], s['vit_bwd'][1]-s['vit_bwd'][0], s['ur_bwd'][1]-s['ur_bwd'][0]))" "$@"; }
run A=0
run HDMOE_BLK6=0
run HDMOE_C6_G=192 HDMOE_B6_G=192
run HDMOE_C6_G=160 HDMOE_B6_G=160 HDMOE_C6S_G=160
run HDMOE_C6_G=128 HDMOE_B6_G=128 HDMOE_C6S_G=128
run HDMOE_C6_G=192 HDMOE_B6_G=192 HDMOE_C6S_G=192
run HDMOE_C6S_G=256
run A=0
