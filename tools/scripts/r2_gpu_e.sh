#!/bin/bash
# GPU box: SSE-compatible mode + CLI tests, default bench with the new pipeline defaults
mkdir -p gpurun_out/profiles
( timeout 1500 python -m pytest tests -m gpu -x -q -k "sse_compatible or cli or eqx" 2>&1 | tail -15 ) > gpurun_out/r2e_pytest.log
( timeout 1500 python bench.py > gpurun_out/r2e_bench.json 2> gpurun_out/r2e_bench.err ); echo "bench rc=$?" >> gpurun_out/r2e_bench.err
tail -8 gpurun_out/r2e_pytest.log; tail -c 400 gpurun_out/r2e_bench.json; tail -2 gpurun_out/r2e_bench.err
