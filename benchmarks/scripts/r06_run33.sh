#!/bin/bash
# round-6 call 33: do VALU phases overlap the MFMA bursts of the other waves of a SIMD?  (benchmarks/micro/mfma_valu_overlap.hip)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run33; O=gpurun_out/r06_run33
hipcc --offload-arch=gfx950 -O3 -w benchmarks/micro/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo | tee $O/mfma_valu_overlap.txt
