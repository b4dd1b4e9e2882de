#!/bin/bash
# GPU box: upload piece size of the uniform streamed plan under the 4-bit wire format (round 6), config 2 end to end, same box back to back
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_piece_ab.txt; : > $O
one() { local label=$1; shift; env "$@" python bench.py --workload cfg2 --steps 30 --warmup 6 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s value %8.1f  flat %8.1f ms/step %8.3f  parity %s' % ('$label', d['value'], d.get('value_flat_arena') or 0, d['ms_per_step'], d['parity_sample']))" >> $O; }
for rep in 1 2 3; do
	one "default" A=1
	for kb in 4096 5600 8192 11200 16800; do one "piece=${kb}KB" KSW2AMD_STREAM_PIECE_KB=$kb; done
done
cat $O
