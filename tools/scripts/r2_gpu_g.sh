#!/bin/bash
# diagnostics: per-chunk timeline of the traceback workloads through the batch entry point
for cfg in "cfg5 6 0" "cfg5 4 1" "cfg3 6 0"; do
	set -- $cfg
	echo "=== WL=$1 THREADS=$2 DBUF=$3" >> gpurun_out/r2g_trace.txt
	KSW2AMD_TRACE=1 KSW2AMD_THREADS=$2 KSW2AMD_DBUF=$3 timeout 300 python bench.py --workload $1 --steps 2 --warmup 2 --no-cpu --no-also 2>> gpurun_out/r2g_trace.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_hbm_resident'], d['ms_per_step'])" >> gpurun_out/r2g_trace.txt
done
tail -60 gpurun_out/r2g_trace.txt
