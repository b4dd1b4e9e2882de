#!/bin/bash
# GPU box: everything the round's profiles/ entries come from.  usage: tools/scripts/round_profiles.sh <tag>
TAG=$1
mkdir -p gpurun_out/profiles
for w in cfg2 cfg3 10k extf; do bash tools/scripts/profile_round.sh $TAG $w 5; done
for w in cfg2 cfg3 10k 10k-cigar cfg4 cfg5 exts extf; do
	n=$(echo $w | tr - _)
	timeout 900 python bench.py --workload $w --steps 10 --warmup 2 --cpu-seconds 8 2> gpurun_out/bench_${TAG}_$w.err | tail -1 > gpurun_out/profiles/${TAG}_bench_$n.json
done
timeout 600 python bench.py --workload cfg2 --approx --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 > gpurun_out/profiles/${TAG}_bench_cfg2_approx.json
timeout 600 python bench.py --workload 10k --approx --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 > gpurun_out/profiles/${TAG}_bench_10k_approx.json
timeout 900 python tools/scripts/ragged_probe.py 3 > gpurun_out/profiles/${TAG}_ragged_probe.txt 2>&1
timeout 600 python tools/scripts/extf_classes.py > gpurun_out/profiles/${TAG}_extf_classes.txt 2>&1
timeout 600 python tools/scripts/exts_classes.py > gpurun_out/profiles/${TAG}_exts_classes.txt 2>&1
ls -la gpurun_out/profiles | tail -40
