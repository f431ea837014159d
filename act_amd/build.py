"""Build libact_hip.so (all HIP kernels + the C ABI of include/act_hip.h) for gfx950 with hipcc.

In-tree build: objects under act_amd/csrc/_obj, library at act_amd/lib/libact_hip.so (git-ignored,
travels to the GPU box with the snapshot).  hipcc cross-compiles without a GPU.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libact_hip.so")
ARCH = "gfx950"

COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-Wno-unused-value", "-DNDEBUG"]
# bit-exact index kernels: never contract a*b+c
PER_FILE = {"point_ops.hip": ["-ffp-contract=off"], "chamfer.hip": ["-ffp-contract=off"]}


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0)")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _sig(path, flags):
    h = hashlib.sha1()
    for p in [path, os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "act_hip.h")] + \
            [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def build(verbose=False, force=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    cc = hipcc()
    objs, jobs = [], []
    for src in sources():
        flags = COMMON + PER_FILE.get(src, []) + os.environ.get("ACT_HIPCC_EXTRA", "").split()   # dev builds only (e.g. -DACT_ATTN_DIAG)
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ, src[:-4] + ".o")
        sigf = obj + ".sig"
        sig = _sig(path, flags)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(sigf) and open(sigf).read() == sig:
            continue
        jobs.append((src, [cc] + flags + ["-c", path, "-o", obj], sigf, sig))

    def run(job):
        src, cmd, sigf, sig = job
        if verbose:
            print("[act_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(sigf, "w") as f:
            f.write(sig)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[act_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
