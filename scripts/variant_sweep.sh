#!/bin/bash
# builds libzlhip.so with each given set of extra hipcc flags and benches it: usage variant_sweep.sh "<bench args>" "<flags1>" "<flags2>" ...
BARGS=$1; shift
for fl in "$@"; do
  ZL_EXTRA_HIPCC_FLAGS="$fl" python3 -c "from libzl_amd import build; build.build_engine(force=True)" || exit 1
  echo "### flags: $fl"
  python3 bench.py $BARGS --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  value=%.3e ms/step=%.3f k2=%.3f ms %.0f GB/s (%.1f%%) k1=%.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['achieved'], 100*r['frac'], r['other_kernels_ms']['zl_k1_plan+k0']))"
done
python3 -c "from libzl_amd import build; build.build_engine(force=True)"
