#!/bin/bash
for v in 0 1 2 3; do
  echo "### variant $v (1=no gather, 2=no mix)"
  ZLHIP_DEBUG_VARIANT=$v python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline $@ 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  k2=%.3f ms k1=%.3f' % (r['avg_launch_ms'], r['other_kernels_ms']['zl_k1_plan+k0']))"
done
