#!/bin/bash
# round-6 call 34: which VALU instruction kinds steal matrix-pipe time from the f32 MFMA (benchmarks/micro/mfma_valu_kinds.hip)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_run34; O=gpurun_out/r06_run34
hipcc --offload-arch=gfx950 -O3 -w benchmarks/micro/mfma_valu_kinds.hip -o /tmp/mvk && /tmp/mvk | tee $O/mfma_valu_kinds.txt
