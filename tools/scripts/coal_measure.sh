# unchanged single-pair callers through the coalescer (GPU box):  bash tools/scripts/coal_measure.sh
cd $GRAFT_REPO_ROOT
B="timeout 120 ./tools/coalesce-bench"
echo "default (slots 4, window 200)"; $B 64 2000 512 64 0; $B 64 2000 512 64 0
for s in 1 2 3; do echo "slots=$s"; KSW2AMD_COALESCE_SLOTS=$s $B 64 2000 512 64 0; done
for w in 0 50 1000; do echo "window $w"; KSW2AMD_COALESCE_WINDOW_US=$w $B 64 2000 512 64 0; done
echo "plain streams"; KSW2AMD_COALESCE_PLAIN_STREAMS=1 $B 64 2000 512 64 0
echo "PK_FIRST=0"; KSW2AMD_PK_FIRST=0 $B 64 2000 512 64 0
for t in 8 16 32 128 256; do echo "$t threads"; $B $t 1500 512 64 0; done
echo cigar; $B 64 1000 512 64 1
echo "1 thread"; $B 1 2000 512 64 0; $B 1 1000 512 64 1
echo trace; KSW2AMD_TRACE=1 $B 64 300 512 64 0 2>&1 | grep -v "cache miss" | sed -n 600,608p
