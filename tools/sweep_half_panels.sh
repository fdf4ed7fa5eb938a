#!/bin/bash
# A/B of the half-tile row panels of the persistent GEMM launch (0 = full tiles only; unset = the model's choice)
cd "$(dirname "$0")/.."
for shape in "fc1:1:12800:3072:768" "qkv:0:12800:2304:768" "fc1L:1:65792:4096:1024" "tfc1:1:19712:2048:512"; do
  for ph in ${PHS:-0 4 6 8 10 12 16 24}; do
    echo -n "ph=$ph  "
    SHAPES=$shape MMR_GEMM_HALF_PANELS=$ph python tools/time_gemm.py 2>&1 | grep TFLOP
  done
done
