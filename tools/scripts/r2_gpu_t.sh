#!/bin/bash
# A/B: target-code planes in LDS (three or four wavefronts per SIMD) against the register form, headline and config 2
for rep in 1 2; do
for v in 0 1; do
	for w in 10k cfg2 10k-n1024; do
		KSW2AMD_LDSCODES=$v timeout 600 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu --resident-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rep$rep ldscodes=$v $w resident', d['roofline']['kernel_gcups'], d['roofline']['kernel_ms'])"
	done
done
done 2>&1 | tee gpurun_out/r2t_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "10k or cfg2 or golden or ragged or fuzz" 2>&1 | tail -3
