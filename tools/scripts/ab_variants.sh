#!/bin/bash
# GPU box: bench every workload with each library variant under build_ab/ (same box, back to back), twice.
for rep in 1 2; do
for v in 00 11 01; do
	cp build_ab/lib_$v.so ksw2_amd/libksw2_amd.so
	for w in cfg2 cfg3 10k 10k-cigar cfg4 cfg5 exts extf; do
		timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep$rep lib_$v', '$w', d['value'], d['roofline']['kernel_ms'])"
	done
done
done
