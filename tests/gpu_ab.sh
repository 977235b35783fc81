#!/bin/bash
# A/B of library variants on the GPU box: tests/gpu_ab.sh <script.py> [args] -- runs it once per
# gadget-leicester_amd/variants/libghip_*.so (alternating twice), prints the last line of each
for rep in 1 2; do
  for so in gadget-leicester_amd/variants/libghip_*.so; do
    echo -n "$(basename $so) : "
    GHIP_LIBGHIP=$PWD/$so timeout -k 10 120 python "$@" 2>&1 | tail -1
  done
done
