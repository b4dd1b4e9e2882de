#!/bin/bash
# A/B of the E-chain variants (build_ab/lib_c{0,1,2,3}.so): resident kernel rate per workload, same box, back to back, twice
cp ksw2_amd/libksw2_amd.so /tmp/keep.so
for rep in 1 2; do
for v in c0 c1 c2 c3; do
	cp build_ab/lib_$v.so ksw2_amd/libksw2_amd.so
	for w in 10k cfg2 10k-cigar cfg4-so cfg4; do
		timeout 600 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu --resident-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rep$rep lib_$v $w resident', d['roofline']['kernel_gcups'], d['roofline']['kernel_ms'])"
	done
done
done 2>&1 | tee gpurun_out/r2s_ab.txt
cp /tmp/keep.so ksw2_amd/libksw2_amd.so
