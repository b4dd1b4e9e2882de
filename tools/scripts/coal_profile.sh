cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_coal1 -o c1 -- ./tools/coalesce-bench 1 300 512 64 0 > gpurun_out/prof_coal1.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_coal64 -o c64 -- ./tools/coalesce-bench 64 300 512 64 0 > gpurun_out/prof_coal64.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_coal64c -o c64c -- ./tools/coalesce-bench 64 300 512 64 1 > gpurun_out/prof_coal64c.log 2>&1
for d in prof_coal1 prof_coal64 prof_coal64c; do echo == $d; find gpurun_out/$d -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200; done
nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)"
