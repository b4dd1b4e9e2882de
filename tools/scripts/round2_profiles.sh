#!/bin/bash
# GPU box: everything the round-2 entries of profiles/ come from.  usage: tools/scripts/round2_profiles.sh <tag>
TAG=$1
mkdir -p gpurun_out/profiles
for w in 10k 10k-cigar cfg2 cfg3 cfg4 cfg5 exts extf; do bash tools/scripts/profile_round.sh $TAG $w 3 > gpurun_out/prof_${TAG}_$w.log 2>&1; done
for w in 10k 10k-cigar cfg2 cfg3 cfg4 cfg5 exts extf; do
	n=$(echo $w | tr - _)
	timeout 900 python bench.py --workload $w --steps 10 --warmup 2 --cpu-seconds 8 --no-also 2> gpurun_out/bench_${TAG}_$w.err | tail -1 > gpurun_out/profiles/${TAG}_bench_$n.json
done
timeout 600 python bench.py --workload 10k --approx --steps 10 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 > gpurun_out/profiles/${TAG}_bench_10k_approx.json
timeout 600 python bench.py --workload cfg2 --sse-compat --pairs 16384 --steps 5 --warmup 1 --no-cpu --no-also 2>/dev/null | tail -1 > gpurun_out/profiles/${TAG}_bench_cfg2_sse_compat.json
timeout 600 python bench.py --workload 10k --sse-compat --pairs 1024 --steps 3 --warmup 1 --no-cpu --no-also 2>/dev/null | tail -1 > gpurun_out/profiles/${TAG}_bench_10k_sse_compat.json
KSW2AMD_NO_PKMP=1 timeout 600 python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu --resident-only 2>/dev/null | tail -1 > gpurun_out/profiles/${TAG}_bench_cfg4_int32_resident.json
for c in 0 1; do timeout 300 tools/coalesce-bench 64 2000 512 64 $c; done > gpurun_out/profiles/${TAG}_coalesce.txt 2>&1
timeout 300 tools/coalesce-bench 1 2000 512 64 0 >> gpurun_out/profiles/${TAG}_coalesce.txt 2>&1
ls -la gpurun_out/profiles | grep $TAG
for f in gpurun_out/profiles/${TAG}_bench_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read())
    print('$f'.split('/')[-1], d.get('value'), d.get('value_hbm_resident', d.get('roofline',{}).get('kernel_gcups')), d.get('roofline',{}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))
except Exception as e: print('$f', 'ERR', e)
"; done
