#!/bin/bash
# find the crashing GPU test
( timeout 1800 python -X faulthandler -m pytest tests -m gpu -x -v 2>&1 | grep -v "^  File\|^$" | head -150 ) > gpurun_out/r2i_pytest.log
tail -40 gpurun_out/r2i_pytest.log
