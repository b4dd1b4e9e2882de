#!/bin/bash
# GPU box: single-pair calls of the ksw2-named entry points through libksw2_amd.so, device path (coalesced when concurrent) against
# the opt-in host path for tiny calls (KSW2AMD_SMALL_CELLS); tools/coalesce-bench checks every result against the batch entry point.
#   usage: tools/scripts/small_call_probe.sh > profiles/<file>
for cig in 0 1; do
for len in 100 250 512 1000 2048; do
  calls=2000; [ $len -ge 1000 ] && calls=500
  for t in 1 64; do
    for small in 0 1000000000; do
      echo -n "threads=$t len=$len band=64 cigar=$cig small_cells=$small  "
      KSW2AMD_SMALL_CELLS=$small timeout 300 tools/coalesce-bench $t $calls $len 64 $cig
    done
  done
done
done
