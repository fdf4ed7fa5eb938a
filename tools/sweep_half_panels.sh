#!/bin/bash
# A/B of the mixed half/full GEMM launch: per-shape time for several counts of half-tile row panels (0 = plain launch)
cd "$(dirname "$0")/.."
for shape in "fc1:1:12800:3072:768" "qkv:0:12800:2304:768" "fc1L:1:65792:4096:1024" "qkvL:0:65792:3072:1024"; do
  for ph in ${PHS:-0 4 6 8 10 12 16 24}; do
    echo -n "ph=$ph  "
    SHAPES=$shape MMR_GEMM_HALF_PANELS=$ph python tools/time_gemm.py 2>&1 | grep TFLOP
  done
done
