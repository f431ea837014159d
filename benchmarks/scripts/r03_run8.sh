cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dense.py -x -q -k "candidate or autotuner or gemm" 2>&1 | tail -3
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import act_amd.kernels as K
from benchmarks.gemm_bench import timeit
for (M, N, Kd) in ((8192, 768, 3072), (8192, 768, 768), (16384, 768, 3072), (16384, 768, 768), (8192, 1536, 768), (8192, 2304, 768), (8192, 3072, 768), (16384, 2304, 768), (16384, 3072, 768), (8192, 384, 384), (8192, 384, 1536), (1792, 384, 1536), (1792, 384, 384), (1792, 1152, 384), (1792, 1536, 384), (8192, 1152, 384), (8192, 1536, 384)):
    a = torch.randn(M, Kd, device="cuda"); b = torch.randn(N, Kd, device="cuda")
    tr = []
    best, bt = K.gemm_tune(a, b, True, True, M, N, Kd, K.workspace(a.device), reps=20, rounds=3, trace=tr)
    fl = 2.0 * M * N * Kd
    tr.sort(key=lambda x: x[2])
    t19 = [(s, round(fl / ms / 1e9, 1)) for t, s, ms in tr if t == 19]
    print(f"nt {M}x{N}x{Kd}: best {best} {bt*1e3:.1f} us {fl/bt/1e9:.1f} TF | shipped {K._GEMM_TABLE.get((1,1,M,N,Kd))} | tile 19: {t19} | top3 {[(t, s, round(fl/ms/1e9,1)) for t, s, ms in tr[:3]]}", flush=True)
PY
