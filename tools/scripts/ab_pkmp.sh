#!/bin/bash
# GPU box: the packed generation-serial class (config 4) under library variants in build_ab/ (K2A_PKMP_WAVES = 3 / 4 / 5, three
# workgroups per CU), resident kernel time, same box, back to back, twice.  The product library is put back at the end.
set -e
KEEP=$(mktemp /tmp/lib_keep.XXXXXX.so)
cp ksw2_amd/libksw2_amd.so "$KEEP"
trap 'cp "$KEEP" ksw2_amd/libksw2_amd.so; rm -f "$KEEP"' EXIT      # an interrupt or a timeout mid-loop must not leave a variant installed as the product
for rep in 1 2; do
for v in base occ3 w5 w3; do
	[ -f build_ab/lib_$v.so ] || { echo "build_ab/lib_$v.so is missing" >&2; exit 1; }
	cp build_ab/lib_$v.so ksw2_amd/libksw2_amd.so
	for w in cfg4 cfg4-so; do
		timeout 900 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu --resident-only 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('rep$rep lib_$v', '$w', 'resident', r['kernel_gcups'], r['kernel_ms'], r['fill_kernel_ms'])"
	done
done
done
