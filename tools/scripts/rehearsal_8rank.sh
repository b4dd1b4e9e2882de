#!/bin/bash
# GPU box: the only 8-GPU rehearsal a 1-GPU box allows -- N gloo ranks sharing ONE MI355X and one host (bench.py's KSW2_BENCH_BACKEND=gloo
# KSW2_BENCH_ONE_DEVICE=1 hooks), against one rank with the whole batch.  usage: rehearsal_8rank.sh [out.json]
OUT=${1:-gpurun_out/rehearsal_8rank.json}
export KSW2_BENCH_BACKEND=gloo KSW2_BENCH_ONE_DEVICE=1 KSW2AMD_MAX_BYTES=$((28*1024*1024*1024))
run1() { python3 bench.py --workload $1 --steps $2 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1; }
runN() { python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 8 --workload $1 --pairs $3 --steps $2 --warmup 2 --no-cpu --no-also 2>/dev/null | tail -1; }
{
echo '{"what": "N gloo ranks sharing ONE MI355X and one host against one rank with the whole batch (tools/scripts/rehearsal_8rank.sh): value = all ranks cells over the slowest rank time", "runs": ['
echo '{"name": "one_rank_10k", "line": '; run1 10k 5; echo '},'
echo '{"name": "eight_ranks_10k", "line": '; runN 10k 5 6144; echo '},'
echo '{"name": "one_rank_cfg2", "line": '; run1 cfg2 20; echo '},'
echo '{"name": "eight_ranks_cfg2", "line": '; runN cfg2 20 8192; echo '},'
echo '{"name": "one_rank_cfg3", "line": '; run1 cfg3 5; echo '},'
echo '{"name": "eight_ranks_cfg3", "line": '; runN cfg3 5 2048; echo '}'
echo ']}'
} > $OUT
python3 - $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for r in d["runs"]:
    l = r["line"]
    print("%-18s value %8.1f GCUPS  ms/step %9.3f  n_gpus %d  parity %s" % (r["name"], l["value"], l["ms_per_step"], l["n_gpus"], l.get("parity_sample")))
PY
