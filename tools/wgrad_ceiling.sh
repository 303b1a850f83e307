#!/bin/bash
# The weight-gradient GEMM's ceiling, measured on its OWN instruction stream: the shipped kernel with mechanisms compiled out
# (DUDF_WGRAD_DBG bits: 1 no staging loads, 8 no operand split, 16 no LDS fragment reads, 64 no output atomics; wrong results, only the clock is
# read), at the clocks the chip really holds.   bash tools/build_dbg.sh wgdbg<N> wgrad "-DDUDF_WGRAD_DBG=<N>" for N in 73 89 65 64 first.
#   89 = MFMAs + flags only | 73 = + LDS fragment reads (operands RESIDENT: no HBM, no split) | 65 = + split | 64 = + staging loads | - = shipped
R=${GRAFT_REPO_ROOT:-/root/repo}
AB_FLAGS="--no-config3" AB_STEPS=30 bash $R/tools/ab.sh "DUDF_LIB=$R/dbg/libdudf_wgdbg89.so" "DUDF_LIB=$R/dbg/libdudf_wgdbg73.so" "DUDF_LIB=$R/dbg/libdudf_wgdbg65.so" "DUDF_LIB=$R/dbg/libdudf_wgdbg64.so" "-"
