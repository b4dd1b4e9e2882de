#!/bin/bash
# headline end to end with the pipeline trace on, default policy and a few chunk/thread settings
for cfg in "6 128" "8 128" "12 128" "6 64"; do
	set -- $cfg
	echo "=== THREADS=$1 CHUNK_MB=$2"
	KSW2AMD_THREADS=$1 KSW2AMD_CHUNK_MB=$2 timeout 600 python bench.py --workload 10k --steps 8 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('10k e2e', d['value'], 'resident', d['value_hbm_resident'], d['config']['host_pipeline'])"
done 2>&1 | tee gpurun_out/r2v_grid.txt
KSW2AMD_TRACE=1 timeout 600 python bench.py --workload 10k --steps 2 --warmup 1 --no-cpu --no-also 2>&1 | grep "ksw2_amd\]" | tail -60 > gpurun_out/r2v_trace.txt
