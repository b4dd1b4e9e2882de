# kernel + copy timeline of a few end-to-end steps of a workload (GPU box):  bash tools/scripts/e2e_prof.sh [workload]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
W=${1:-cfg2}
timeout 280 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_${W}_e2e -o t -- python3 bench.py --workload $W --steps 6 --warmup 12 --no-cpu --no-also > gpurun_out/prof_${W}_e2e.log 2>&1
ls gpurun_out/prof_${W}_e2e/
