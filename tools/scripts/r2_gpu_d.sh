#!/bin/bash
# chunk policy grid, second pass: headline and the short-pair workloads
for cfg in "10k 256 4 0" "10k 128 3 0" "10k 128 2 0" "10k 192 4 0" "10k 128 6 0" "cfg2 64 4 1" "cfg2 128 4 0" "cfg2 128 8 0" "cfg2 16 8 0" "cfg3 64 4 1" "cfg3 128 4 0" "cfg3 128 8 0" "cfg5 64 4 1" "cfg5 128 4 0" "cfg5 256 4 0" "10k-cigar 64 4 1" "10k-cigar 128 4 0"; do
	set -- $cfg
	echo "WL=$1 CHUNK_MB=$2 THREADS=$3 DBUF=$4" >> gpurun_out/r2d_grid.txt
	KSW2AMD_CHUNK_MB=$2 KSW2AMD_THREADS=$3 KSW2AMD_DBUF=$4 KSW2AMD_RAMP=0 timeout 300 python bench.py --workload $1 --steps 6 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_hbm_resident'], d['ms_per_step'], d['config']['host_pipeline'])" >> gpurun_out/r2d_grid.txt
done
cat gpurun_out/r2d_grid.txt
