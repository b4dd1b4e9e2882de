#!/bin/bash
# GPU box: round-6 check of the target-wildcard rule -- the new GPU test, then the workloads whose kernels gained registers
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_exp2.txt; : > $O
python -m pytest tests/test_gpu_parity.py -q -x -k "target_wildcards or frozen_books or uniform_plans or streamed_plans or solo_kernel or packed_generation_serial or deferred_argmax or headline_kernel" 2>&1 | tail -4 >> $O
one() { local label=$1 wl=$2; shift 2; env "$@" python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %-10s value %8.1f flat %8.1f resident %8.1f kernel_ms %8.3f parity %s' % ('$label', '$wl', d['value'], d.get('value_flat_arena') or 0, d['value_hbm_resident'], d['roofline']['kernel_ms'], d['parity_sample']))" >> $O; }
one default 10k-tN A=1
one TN=0 10k-tN KSW2AMD_TN=0
one default 10k A=1
one default cfg2 A=1
one default cfg2 A=1
one default 10k-n1024 A=1
one default cfg5 A=1
one default cfg3 A=1
one default cfg4 A=1
one default 10k-cigar A=1
cat $O
