#!/bin/bash
# GPU box, round 2 second pass: default bench (double-buffered chunk pipeline), host pipeline tests
mkdir -p gpurun_out/profiles
( timeout 1500 python bench.py > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err ); echo "bench rc=$?" >> gpurun_out/r2b_bench.err
( timeout 900 python -m pytest tests -m gpu -x -q -k "pipeline or pool or coalesc or thread" 2>&1 | tail -8 ) > gpurun_out/r2b_pytest.log
for c in 0 1; do timeout 300 tools/coalesce-bench 64 2000 512 64 $c; done > gpurun_out/r2b_coalesce.txt 2>&1
tail -3 gpurun_out/r2b_pytest.log; tail -c 600 gpurun_out/r2b_bench.json; tail -3 gpurun_out/r2b_bench.err; cat gpurun_out/r2b_coalesce.txt
