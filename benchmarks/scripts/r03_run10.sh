cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import act_amd.kernels as K
for (M, N, Kd) in ((256, 128, 262144), (256, 128, 1048576), (256, 128, 131072), (256, 128, 65536), (512, 256, 131072), (512, 256, 65536), (384, 512, 131072), (384, 512, 65536)):
    a = torch.randn(Kd, M, device="cuda"); b = torch.randn(Kd, N, device="cuda")
    tr = []
    best, bt = K.gemm_tune(a, b, False, False, M, N, Kd, K.workspace(a.device), reps=5, rounds=3, trace=tr)
    fl = 2.0 * M * N * Kd
    tr.sort(key=lambda x: x[2])
    sh = K._GEMM_TABLE.get((0, 0, M, N, Kd))
    cur = [ms for t, s, ms in tr if (t, s) == tuple(sh)] if sh else []
    print(f"tn {M}x{N}x{Kd}: best {best} {bt*1e3:.1f} us {fl/bt/1e9:.1f} TF | shipped {sh} {cur[0]*1e3 if cur else -1:.1f} us | top4 {[(t, s, round(ms*1e3,1)) for t, s, ms in tr[:4]]}", flush=True)
PY
