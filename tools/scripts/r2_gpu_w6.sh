#!/bin/bash
# packed generation-serial class with 6 instead of 4 wavefronts per task (build_ab/lib_w6.so), config 4 resident + parity
cp ksw2_amd/libksw2_amd.so /tmp/keep.so
for rep in 1 2; do
for v in w4 w6; do
	cp build_ab/lib_$v.so ksw2_amd/libksw2_amd.so
	for w in cfg4 cfg4-so; do
		timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu --resident-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rep$rep lib_$v $w resident', d['roofline']['kernel_gcups'], d['roofline']['kernel_ms'], d['roofline']['fill_kernel_ms'])"
	done
done
done 2>&1 | tee gpurun_out/r2_w6_ab.txt
cp build_ab/lib_w6.so ksw2_amd/libksw2_amd.so
timeout 900 python -m pytest tests -m gpu -x -q -k "packed_generation or mt_pair or cfg4 or 50k" 2>&1 | tail -3
cp /tmp/keep.so ksw2_amd/libksw2_amd.so
