cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dense.py -x -q -k "candidate or autotuner or 32_deep" 2>&1 | tail -3
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import act_amd.kernels as K
from benchmarks.gemm_bench import timeit
for (M, N, Kd) in ((8192, 3072, 768), (8192, 768, 3072), (8192, 2304, 768), (8192, 1536, 768), (8192, 768, 768), (8192, 8192, 2304), (16384, 3072, 768), (16384, 768, 3072), (16384, 2304, 768), (16384, 768, 768), (8192, 384, 1536), (8192, 1536, 384), (8192, 1152, 384)):
    a = torch.randn(M, Kd, device="cuda"); b = torch.randn(N, Kd, device="cuda"); out = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * N * Kd
    line = f"nt {M}x{N}x{Kd}:"
    for cfg in ((10, 1), (20, 1), (11, 1), (21, 1), (10, 2), (20, 2), (17, 1), (17, 2)):
        try:
            t = min(timeit(lambda: K.gemm(a, b, True, True, out=out, cfg=cfg), 30) for _ in range(3))
            line += f"  {cfg}: {fl/t/1e9:6.1f}"
        except Exception:
            line += f"  {cfg}:    n/a"
    print(line, "| shipped", K._GEMM_TABLE.get((1, 1, M, N, Kd)), flush=True)
PY
