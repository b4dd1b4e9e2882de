#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_exp3.txt; : > $O
python -m pytest tests/test_gpu_parity.py -q -x -k "target_wildcards or frozen_books or uniform_plans or streamed_plans or solo_kernel or packed_generation_serial or deferred_argmax or headline_kernel or flat_batches" 2>&1 | tail -4 >> $O
bash tools/scripts/ab_libs.sh "r6base tn_body" "10k cfg2 10k-n1024 cfg4 10k-cigar cfg3" 2 gpurun_out/r6_ab_tn_body.txt > /dev/null
cat gpurun_out/r6_ab_tn_body.txt >> $O
for w in 10k-tN 10k; do python bench.py --workload $w --steps 8 --warmup 3 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product    %-10s value %8.1f flat %8.1f resident %8.1f kernel_ms %9.4f parity %s' % ('$w', d['value'], d.get('value_flat_arena') or 0, d['value_hbm_resident'], d['roofline']['kernel_ms'], d['parity_sample']))" >> $O; done
cat $O
