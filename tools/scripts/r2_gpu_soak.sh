#!/bin/bash
# longer randomised soak of the final state: short and long modes, new seeds; pipeline consistency with another seed
( env -u KSW2AMD_SIMDS timeout 700 python tools/scripts/fuzz_gpu.py 600 20260101 2>&1 | tail -2 ) > gpurun_out/r2_soak.txt
( env -u KSW2AMD_SIMDS timeout 700 python tools/scripts/fuzz_gpu.py 600 20260102 long 2>&1 | tail -2 ) >> gpurun_out/r2_soak.txt
( timeout 900 python tools/scripts/pipeline_consistency.py 23 2>&1 | tail -3 ) >> gpurun_out/r2_soak.txt
cat gpurun_out/r2_soak.txt
