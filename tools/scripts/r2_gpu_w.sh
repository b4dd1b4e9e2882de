#!/bin/bash
# headline with the 12-chunk fix; config 2 / config 3 / 10k-cigar pipeline traces and thread grid
timeout 600 python bench.py --workload 10k --steps 12 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('10k e2e', d['value'], 'resident', d['value_hbm_resident'], d['config']['host_pipeline'])"
timeout 600 python bench.py --workload 10k --approx --steps 12 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('10k approx e2e', d['value'], 'resident', d['value_hbm_resident'], d['config']['host_pipeline'])"
for w in cfg2 cfg3 10k-cigar; do
for t in 6 8 12 16; do
	KSW2AMD_THREADS=$t timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w THREADS=$t e2e', d['value'], 'resident', d['value_hbm_resident'], d['config']['host_pipeline'])"
done
done
for w in cfg2 cfg3; do
echo "=== trace $w"
KSW2AMD_TRACE=1 timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu --no-also 2>&1 | grep "ksw2_amd\]" | tail -40
done
