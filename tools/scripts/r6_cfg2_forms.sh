#!/bin/bash
# GPU box: config 2's kernel forms, resident rate, same box back to back (round 6 re-check of profiles/r3_cfg2_geometry_ab.txt after round 5's one-selector-per-row kernels)
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_cfg2_forms.txt; : > $O
one() { local label=$1; shift; env "$@" python bench.py --workload cfg2 --steps 20 --warmup 3 --no-cpu --resident-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s resident %8.1f kernel_ms %8.4f %s' % ('$label', r['kernel_gcups'], r['kernel_ms'], r.get('kernels')))" >> $O; }
for rep in 1 2; do
one "default (8,18) registers" A=1
one "(8,18) LDS selectors" KSW2AMD_LDSCODES=1
one "(16,8) auto" KSW2AMD_PK_FIRST=1
one "(16,8) registers" KSW2AMD_PK_FIRST=1 KSW2AMD_LDSCODES=0
one "(16,8) LDS selectors" KSW2AMD_PK_FIRST=1 KSW2AMD_LDSCODES=1
done
cat $O
