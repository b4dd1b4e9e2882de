#!/bin/bash
# GPU box: tools/scripts/profile_round.sh for every workload of bench.py's default `also` list + the headline.
#   usage: tools/scripts/profile_all.sh <tag> [workload ...]      -> gpurun_out/profiles/<tag>_<workload>_{kernel_stats.csv,pmc.json}
TAG=${1:-r5p}; shift
WLS=${@:-10k 10k-n1024 10k-cigar cfg2 cfg3 cfg5 cfg4 10k-zdrop 10k-N 10k-tN 10k-generic 10k-approx exts extf extf-w300 extf-w900 10k-ssec 10k-ssec-n4096 10k-ssec-approx 10k-ssec-cigar}
for wl in $WLS; do
	st=3; case $wl in cfg2) st=10;; exts|extf|10k-n1024) st=6;; esac
	t0=$(date +%s)
	bash tools/scripts/profile_round.sh $TAG $wl $st > gpurun_out/prof_${TAG}_${wl}.log 2>&1
	echo "$wl: $(( $(date +%s) - t0 )) s; $(tail -c 200 gpurun_out/prof_${TAG}_${wl}.log | tr '\n' ' ')"
	rm -rf gpurun_out/prof_${TAG}_${wl}          # the raw rocprofv3 output; the summaries are under gpurun_out/profiles/
done
# the round's host-contention rehearsal (8 gloo ranks sharing the one device and host): part of a full pass
# (no workload list given) or with REHEARSAL=1
if [ "${REHEARSAL:-}" = "1" ] || [ -z "${1:-}" ]; then
	bash tools/scripts/rehearsal_8rank.sh gpurun_out/profiles/${TAG}_rehearsal_8rank_one_device.json 2>&1 | tail -7
fi
