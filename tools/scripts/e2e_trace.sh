# end-to-end numbers of the configurations (GPU box):  bash tools/scripts/e2e_trace.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $W $*"; env "$@" timeout 300 python bench.py --workload $W --steps $S --warmup 5 --no-cpu --no-also 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('value_flat_arena'), d.get('value_hbm_resident'), d['ms_per_step'], d.get('parity_sample'))"; }
S=100
for W in cfg2 cfg3; do run A=1; run A=1; done
S=20
for W in 10k-cigar 10k cfg5; do run A=1; done
timeout 120 ./tools/coalesce-bench 64 2000 512 64 0
timeout 120 ./tools/coalesce-bench 1 2000 512 64 0
timeout 600 python -m pytest tests -m gpu -x -q -k "flat or stream or uniform or golden" 2>&1 | tail -2
