#!/bin/bash
# dev tool: sweep two env knobs over bench.py; usage: tools/sweep2.sh "A=1 B=2" "A=3 B=4" ...
for kv in "$@"; do
  ms=$(env $kv timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "$kv ms_per_step=$ms"
done
