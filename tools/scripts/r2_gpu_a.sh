#!/bin/bash
# GPU box, round 2 first pass: parity tests, default bench, threaded-caller bench, SSE agreement
mkdir -p gpurun_out/profiles
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r2a_pytest.log
( timeout 1200 python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err ); echo "bench rc=$?" >> gpurun_out/r2a_bench.err
for c in 0 1; do timeout 300 tools/coalesce-bench 64 2000 512 64 $c; done > gpurun_out/r2a_coalesce.txt 2>&1
KSW2AMD_COALESCE_SLOTS=0 timeout 300 tools/coalesce-bench 64 500 512 64 0 >> gpurun_out/r2a_coalesce.txt 2>&1
timeout 300 tools/coalesce-bench 1 2000 512 64 0 >> gpurun_out/r2a_coalesce.txt 2>&1
timeout 600 python tools/scripts/sse_agreement.py hip gpurun_out/profiles/r2_sse_agreement.json > gpurun_out/r2a_agree.txt 2>&1
tail -5 gpurun_out/r2a_pytest.log; tail -c 1500 gpurun_out/r2a_bench.json; tail -3 gpurun_out/r2a_bench.err; cat gpurun_out/r2a_coalesce.txt
