#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_exp4.txt; : > $O
python -m pytest tests/test_gpu_parity.py -q -x -k "uniform_plans or streamed_plans or sse_compatible or approx or target_wildcards or flat_batch" 2>&1 | tail -4 >> $O
one() { local label=$1 wl=$2; shift 2; env "$@" python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %-16s value %8.1f flat %8.1f resident %8.1f kernel_ms %9.4f parity %s' % ('$label', '$wl', d['value'], d.get('value_flat_arena') or 0, d['value_hbm_resident'], d['roofline']['kernel_ms'], d['parity_sample']))" >> $O; }
for rep in 1 2 3; do
one wire2 cfg2 A=1
one wire4 cfg2 KSW2AMD_WIRE2=0
done
one wire2 10k A=1
one wire4 10k KSW2AMD_WIRE2=0
one wire2 10k-tN A=1
for w in 10k-approx 10k-ssec 10k-ssec-n4096 10k-ssec-approx 10k-ssec-cigar; do
python bench.py --workload $w --steps 4 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product    %-16s value %8.1f flat %8.1f resident %8.1f kernel_ms %9.4f frac %.4f parity %s' % ('$w', d['value'], d.get('value_flat_arena') or 0, d['value_hbm_resident'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['parity_sample']))" >> $O
done
cat $O
