#!/bin/bash
for v in 00 11 01; do
	cp build_ab/lib_$v.so ksw2_amd/libksw2_amd.so
	for w in cfg3 10k 10k-cigar cfg5; do
		KSW2AMD_NO_PK=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nopk lib_$v', '$w', d['value'], d['roofline']['kernel_ms'])"
	done
	KSW2AMD_SOLO=1 timeout 600 python tools/scripts/ragged_probe.py 3 2>&1 | grep "solo" | sed "s/^/solo lib_$v /"
done
