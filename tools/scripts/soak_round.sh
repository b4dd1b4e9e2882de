# Soak of the final state of a round (GPU box, ~25 min):  bash tools/scripts/soak_round.sh > gpurun_out/soak.txt
cd $GRAFT_REPO_ROOT
echo "== fuzz 600 s seed 401"; timeout 700 python tools/scripts/fuzz_gpu.py 600 401 2>&1 | tail -2
echo "== fuzz 300 s seed 402 long reads"; timeout 400 python tools/scripts/fuzz_gpu.py 300 402 long 2>&1 | tail -2
echo "== pipeline consistency seed 9"; timeout 600 python tools/scripts/pipeline_consistency.py 9 2>&1 | tail -3
echo "== streamed plans stress"; timeout 400 python tools/scripts/stream_stress.py 60 2>&1 | tail -8
echo "== coalesce-bench mixed sizes (every result against the batch entry)"
for a in "64 3000 100 16 0" "64 2000 512 64 1" "200 1000 300 32 0" "64 300 3000 200 1" "16 3000 512 64 0"; do timeout 200 ./tools/coalesce-bench $a | cut -c1-210; done
