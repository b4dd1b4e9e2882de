#!/bin/bash
# GPU box: round-6 A/B of existing switches -- (8,18) DEFER on config 2, the target-wildcard workload, per-kernel stats of 10k-tN
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_exp1.txt; : > $O
one() { local label=$1 wl=$2; shift 2; env "$@" python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s %-8s value %8.1f flat %8.1f resident %8.1f kernel_ms %8.3f parity %s' % ('$label', '$wl', d['value'], d.get('value_flat_arena') or 0, d['value_hbm_resident'], d['roofline']['kernel_ms'], d['parity_sample']))" >> $O; }
for rep in 1 2; do
one default cfg2 A=1
one DEFER=1 cfg2 KSW2AMD_DEFER=1
done
one default 10k-tN A=1
one default 10k A=1
cat $O
