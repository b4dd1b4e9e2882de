cd $GRAFT_REPO_ROOT
for s in 411 412 413; do echo "== fuzz 300 s seed $s"; timeout 400 python tools/scripts/fuzz_gpu.py 300 $s 2>&1 | tail -1; done
echo "== fuzz 240 s seed 414 long reads"; timeout 340 python tools/scripts/fuzz_gpu.py 240 414 long 2>&1 | tail -1
echo "== pipeline consistency seed 10"; timeout 600 python tools/scripts/pipeline_consistency.py 10 2>&1 | tail -1
