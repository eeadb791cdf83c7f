for v in "X=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "GPU_MAX_HW_QUEUES=8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1"; do
  echo "== $v"
  env $v timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
