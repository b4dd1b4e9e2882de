cd $GRAFT_REPO_ROOT
timeout 120 ./build_ab/cskp
echo "GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 60 ./build_ab/cskp | grep "events 1, streams non-blocking, copies 1"
B="timeout 120 ./tools/coalesce-bench"
KSW2AMD_COALESCE_WINDOW_US=1000 KSW2AMD_TRACE=1 $B 64 300 512 64 0 2>&1 | grep -v "cache miss" | sed -n 2000,2060p
