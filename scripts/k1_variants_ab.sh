#!/bin/bash
# K1 experiments of round 4, BASELINE config 2 and three smaller N, interleaved rounds in one process:
#   libpgi.so          the default (four wavefronts per workgroup; N = 2000 runs in the hybrid class, 1280 rows in LDS)
#   libpgi_nw8.so      eight wavefronts per workgroup, two workgroups per CU, all 2000 rows in LDS (make libpgi_nw8.so)
#   libpgi_loop.so     the size-class launches as persistent grids (make libpgi_loop.so), PGI_K1_PERSISTENT = 0 / 1
# Usage: scripts/k1_variants_ab.sh > gpurun_out/k1_variants.txt
cd "$(dirname "$0")/.."
echo "== default vs NW = 8 (AB_P = 10000)"
AB_P=10000 AB_ROUNDS=9 python scripts/ab_compare.py libpgi.so libpgi_nw8.so
echo "== default vs loop build, grid = pairs (PGI_K1_PERSISTENT=0)"
AB_P=10000 AB_ROUNDS=9 PGI_K1_PERSISTENT=0 python scripts/ab_compare.py libpgi.so libpgi_loop.so
echo "== default vs loop build, persistent class grids (PGI_K1_PERSISTENT=1)"
AB_P=10000 AB_ROUNDS=9 PGI_K1_PERSISTENT=1 python scripts/ab_compare.py libpgi.so libpgi_loop.so
