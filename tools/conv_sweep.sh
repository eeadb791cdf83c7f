#!/bin/bash
# dev tool: v5 vs v3 forward/backward on the layer shapes of config 2 (run on the GPU box)
for args in "dtype=bf16 rows=512 hw=32 cin=64 cout=64 ks=3" "dtype=bf16 rows=512 hw=32 cin=32 cout=32 ks=3,3,5,5" "dtype=bf16 rows=512 hw=16 cin=64 cout=64 ks=3,3,5,5" "dtype=bf16 rows=512 hw=16 cin=128 cout=64 ks=3,3,5,5" "dtype=bf16 rows=512 hw=32 cin=96 cout=32 ks=3,3,5,5" "dtype=fp32 rows=256 hw=32 cin=64 cout=128 ks=3" "dtype=fp32 rows=256 hw=32 cin=128 cout=128 ks=3" "dtype=fp32 rows=256 hw=32 cin=32 cout=64 ks=3"; do
  echo "== $args"
  python tools/conv_bench.py $args iters=30 | sed -n '2p;4p'
  HDMOE_CONV_V3=1 python tools/conv_bench.py $args iters=30 | sed -n '2p;4p' | sed 's/^/   v3: /'
done
