# same-box comparison of two checkouts of the repository (each with its own built libhdmoe_hip.so):  tools/ab_trees.sh DIR_A DIR_B [rounds]
A=$1; B=$2; n=${3:-3}
one() { (cd $1 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s %.3f ms/step  %.2f steps/s' % (sys.argv[1], d['ms_per_step'], d['value']))" $2); }
for i in $(seq $n); do one $A tree_A ""; one $B tree_B "--no-sampler"; done
