#!/bin/bash
run() { timeout 600 python bench.py --workload $1 --steps 10 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 e2e', d['value'], 'resident', d['value_hbm_resident'], d['config']['host_pipeline'])"; }
for t in 4 6 8 12; do for db in 0 1; do
	KSW2AMD_LONG_MS=20 KSW2AMD_THREADS=$t KSW2AMD_DBUF=$db run cfg3 "LONG_MS=20 THREADS=$t DBUF=$db"
done; done
for t in 6 8 12; do KSW2AMD_THREADS=$t KSW2AMD_DBUF=1 run cfg2 "THREADS=$t DBUF=1"; done
for t in 2 4; do KSW2AMD_POOL_MIN=512 KSW2AMD_THREADS=$t run 10k-cigar "forced pool THREADS=$t"; done
KSW2AMD_LONG_MS=20 run cfg5 "LONG_MS=20"
run cfg5 "default"
