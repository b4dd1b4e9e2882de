#!/bin/bash
# GPU box: bench every workload with each library variant under build_ab/ (same box, back to back), twice.
# The variants are builds of ksw2_shim_hip.hip with the scalar wavefront index (k2a_wave_id<UNIFORM>) forced on or off per
# kernel family; in the build container, e.g. for "fill kernels vector, one-task-per-wavefront kernels scalar" (lib_01):
#   sed 's/k2a_wave_id<(C <= 8)>()/k2a_wave_id<false>()/' ksw2_amd/csrc/ksw2_shim_hip.hip > ksw2_amd/csrc/_v.hip
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c ksw2_amd/csrc/_v.hip -o /tmp/v.o
#   hipcc --offload-arch=gfx950 -shared -o build_ab/lib_01.so ksw2_amd/csrc/ksw2_host.o /tmp/v.o -ldl
# (build_ab/ travels to the GPU box with the snapshot and is git-ignored).
for rep in 1 2; do
for v in 00 11 01; do
	cp build_ab/lib_$v.so ksw2_amd/libksw2_amd.so
	for w in cfg2 cfg3 10k 10k-cigar cfg4 cfg5 exts extf; do
		timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep$rep lib_$v', '$w', d['value'], d['roofline']['kernel_ms'])"
	done
done
done
