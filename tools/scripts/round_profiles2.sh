#!/bin/bash
# GPU box: refresh of the entries that changed after round_profiles.sh <tag> (same outputs, fewer workloads).
TAG=$1
mkdir -p gpurun_out/profiles
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/scripts/profile_round.sh $TAG cfg5 3 > /dev/null 2>&1
for w in cfg2 cfg3 10k 10k-cigar cfg4 cfg5 exts extf; do
	n=$(echo $w | tr - _)
	timeout 900 python bench.py --workload $w --steps 10 --warmup 2 --cpu-seconds 8 2> gpurun_out/bench_${TAG}_$w.err | tail -1 > gpurun_out/profiles/${TAG}_bench_$n.json
done
timeout 600 python bench.py --workload cfg5 --approx --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1 > gpurun_out/profiles/${TAG}_bench_cfg5_approx.json
timeout 600 python tools/scripts/mp_probe.py > gpurun_out/profiles/${TAG}_mp_probe.txt 2>&1
ls gpurun_out/profiles | grep $TAG
