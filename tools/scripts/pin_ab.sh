#!/bin/bash
# GPU box: pool workers pinned to the GPU's NUMA node (default) against unpinned (KSW2AMD_PIN=0), end to end, same box back to back.
one() { local label=$1 wl=$2 st=$3; shift 3; env "$@" python bench.py --workload $wl --steps $st --warmup 4 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s %-10s value %8.1f  flat %8.1f  ms/step %8.3f  parity %s' % ('$label', '$wl', d['value'], d['value_flat_arena'] or 0, d['ms_per_step'], d['parity_sample']))"; }
for rep in 1 2; do
	for wl in cfg2 cfg3 10k-cigar; do one pin=1 $wl 12 KSW2AMD_PIN=1; one pin=0 $wl 12 KSW2AMD_PIN=0; done
	one pin=1 10k 8 KSW2AMD_PIN=1; one pin=0 10k 8 KSW2AMD_PIN=0
done
one pin=1 cfg5 4 KSW2AMD_PIN=1; one pin=0 cfg5 4 KSW2AMD_PIN=0
cat /sys/bus/pci/devices/*/local_cpulist 2>/dev/null | sort | uniq -c | head -5
