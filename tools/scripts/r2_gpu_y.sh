#!/bin/bash
KSW2AMD_TRACE=2 timeout 600 python bench.py --workload cfg2 --steps 3 --warmup 2 --no-cpu --no-also 2>&1 | grep "ksw2_amd\]" | tail -26
