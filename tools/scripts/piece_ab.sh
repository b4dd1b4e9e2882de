#!/bin/bash
# GPU box: upload piece size of the streamed uniform plan (KSW2AMD_STREAM_PIECE_KB), config 2 end to end, same box back to back.
one() { local label=$1; shift; env "$@" python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-18s value %8.1f  ms/step %8.3f  parity %s' % ('$label', d['value'], d['ms_per_step'], d['parity_sample']))"; }
for rep in 1 2; do
	one "default" KSW2AMD_UNIFORM=1
	for kb in 4096 8192 12288 16384 33554; do one "piece=${kb}KB" KSW2AMD_UNIFORM=1 KSW2AMD_STREAM_PIECE_KB=$kb; done
	one "threads=12 8MB" KSW2AMD_UNIFORM=1 KSW2AMD_STREAM_PIECE_KB=8192 KSW2AMD_THREADS=12
	one "chunks" KSW2AMD_UNIFORM=0
done
