#!/bin/bash
# diagnostics: steady-state per-chunk timeline of cfg3 and 10k-cigar, a few pool configurations
for cfg in "cfg3 6 128" "cfg3 4 64" "10k-cigar 6 128" "cfg5 6 128"; do
	set -- $cfg
	echo "=== WL=$1 THREADS=$2 CHUNK=$3" >> gpurun_out/r2k_trace.txt
	KSW2AMD_TRACE=1 KSW2AMD_THREADS=$2 KSW2AMD_CHUNK_MB=$3 timeout 300 python bench.py --workload $1 --steps 3 --warmup 2 --no-cpu --no-also 2> gpurun_out/r2k_err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_hbm_resident'], d['ms_per_step'], d['config']['host_pipeline'])" >> gpurun_out/r2k_trace.txt
	tail -40 gpurun_out/r2k_err.txt >> gpurun_out/r2k_trace.txt
done
cat gpurun_out/r2k_trace.txt | cut -c1-150
