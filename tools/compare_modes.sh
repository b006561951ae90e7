#!/usr/bin/env bash
# MF (mode 0) vs vector-ALU streaming (mode 2) on the BASELINE shapes; two rounds each
for shape in "--N 256 --D 768 --Q 8 --B 1024" "--N 197 --D 768 --Q 8 --B 1024" "--N 196 --D 1024 --Q 8 --B 768" "--N 256 --D 1152 --Q 8 --B 512" "--N 196 --D 384 --Q 1 --B 2048" "--N 256 --D 768 --Q 16 --B 1024"; do
  for round in 1 2; do for m in ${MODES:-0 2}; do
    echo -n "[$shape] mode=$m: "; EP_POOL_MODE=$m python tools/pool_microbench.py --bwd $shape 2>/dev/null | cut -c1-95
  done; done
done
