#!/bin/bash
# dev tool: sweep an env knob over bench.py; usage: tools/sweep.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  ms=$(env $var=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "$var=$v ms_per_step=$ms"
done
