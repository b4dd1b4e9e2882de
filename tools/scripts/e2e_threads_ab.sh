#!/bin/bash
# GPU box: end-to-end rate of the batch entry points under worker-pool sizes (KSW2AMD_THREADS) and the forced single streamed plan,
# same box back to back.  usage: tools/scripts/e2e_threads_ab.sh  > gpurun_out/<tag>/e2e_threads_ab.txt
run() { # label, workload, env...
	local label=$1 wl=$2; shift 2
	env "$@" python bench.py --workload $wl --steps 12 --warmup 4 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s %-6s value %8.1f  flat %8.1f  resident %8.1f  ms/step %7.3f' % ('$label', '$wl', d['value'], d['value_flat_arena'] or 0, d['value_hbm_resident'], d['ms_per_step']))"
}
for rep in 1 2; do
	for t in 6 8 12 16; do run "threads=$t" cfg2 KSW2AMD_THREADS=$t; done
	run "stream=1 threads=6" cfg2 KSW2AMD_STREAM=1
	run "stream=1 threads=12" cfg2 KSW2AMD_STREAM=1 KSW2AMD_THREADS=12
	for t in 6 8 12; do run "threads=$t" cfg3 KSW2AMD_THREADS=$t; done
	for t in 6 8 12; do run "threads=$t" 10k-cigar KSW2AMD_THREADS=$t; done
done
for t in 6 8 12; do run "threads=$t" cfg5 KSW2AMD_THREADS=$t; done
