#!/bin/bash
# GPU box: resident + end-to-end rates of bench.py workloads under library variants in build_ab/ (selected through KSW2AMD_LIB: the
# product .so is never touched), same box, back to back, REPS times.   usage: ab_libs.sh "<variants>" "<workloads>" [reps] [out]
cd "$(dirname "$0")/../.."
VARS=${1:-"r6base tn_mid"}; WLS=${2:-"10k cfg2 10k-n1024 cfg4 10k-cigar cfg3 cfg5"}; REPS=${3:-2}; O=${4:-gpurun_out/ab_libs.txt}
: > $O
for rep in $(seq 1 $REPS); do
for w in $WLS; do
for v in $VARS; do
	[ -f build_ab/lib_$v.so ] || { echo "build_ab/lib_$v.so is missing" >&2; exit 1; }
	KSW2AMD_LIB=$PWD/build_ab/lib_$v.so timeout 600 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep$rep %-10s %-10s value %8.1f resident %8.1f kernel_ms %9.4f parity %s' % ('$v', '$w', d['value'], d['value_hbm_resident'], d['roofline']['kernel_ms'], d['parity_sample']))" >> $O
done
done
done
cat $O
