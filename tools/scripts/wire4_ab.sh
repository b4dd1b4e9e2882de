#!/bin/bash
# GPU box: uniform plans with the 4-bit wire format (two residue codes per byte in staging and upload, expanded on the device) on / off
# (KSW2AMD_WIRE4), end to end, same box back to back.
one() { # label workload steps env...
	local label=$1 wl=$2 st=$3; shift 3
	env "$@" python bench.py --workload $wl --steps $st --warmup 5 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s %-9s value %8.1f  flat %8.1f  resident %8.1f  ms/step %8.3f  parity %s' % ('$label', '$wl', d['value'], d['value_flat_arena'] or 0, d['value_hbm_resident'], d['ms_per_step'], d['parity_sample']))"
}
for rep in 1 2 3; do one wire4=1 cfg2 20 KSW2AMD_WIRE4=1; one wire4=0 cfg2 20 KSW2AMD_WIRE4=0; done
for rep in 1 2; do one wire4=1 10k 10 KSW2AMD_WIRE4=1; one wire4=0 10k 10 KSW2AMD_WIRE4=0; done
for rep in 1; do one wire4=1 10k-N 5 KSW2AMD_WIRE4=1; one wire4=0 10k-N 5 KSW2AMD_WIRE4=0; done
