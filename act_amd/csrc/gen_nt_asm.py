#!/usr/bin/env python3
"""Generator of the hand-scheduled NT main loop (gfx950 assembly inside one inline-asm statement) -> gemm_nt_asm_loop.h

Why a generator: the loop is ~1,500 instructions per tile shape whose ORDER is the whole point (which MFMA gap a ds_read / ds_write /
global_load / s_waitcnt / s_barrier sits in); the schedule is described here once and emitted for every tile shape and loop phase.

The loop (per wave; 2 x 2 waves per workgroup, wave tile 16 TM x 16 TN, K tile 32 = two 16-deep sub-tiles `kg`):

    registers   acc[TM][TN] (4 each) | fragment set 0 (A: TM x float4, B: TN x float4) | set 1 | staging (TM + TN float4) | offsets / LDS bases
    LDS         stage s (0 / 1): A kg0 | A kg1 | B kg0 | B kg1, each [rows][16 k] with the XOR chunk swizzle of gemm_nt16_kernel.h
    iteration t phase A   4 NM MFMAs on set 0 (tile t, kg0); in their gaps: ds_read set 1 <- stage s, kg1
                phase B   4 NM MFMAs on set 1 (tile t, kg1); in their gaps: vmcnt(0), ds_write staging -> stage s^1 (tile t+1),
                          global_load tile t+2 -> staging, lgkmcnt(0) + s_barrier TAIL MFMAs before the end, ds_read set 0 <- stage s^1, kg0
    one barrier per 8 NM MFMAs, every memory instruction in an MFMA gap, no s_waitcnt that can stall in steady state.

Products and their order per accumulator are those of the 16-deep compiler loop (sgemm_nt16_kernel): results are bit-identical.
Run: python gen_nt_asm.py   (writes gemm_nt_asm_loop.h next to this file; the header is committed, this script documents it)
"""
import os
import sys

S_A, S_B, S_CNT = 60, 62, 64          # pinned SGPRs: A base (pair), B base (pair), K-tile count


class Regs:
    def __init__(self, TM, TN, affine=False):
        self.TM, self.TN = TM, TN
        n = 0
        self.acc = n; n += TM * TN * 4
        self.fa = [0, 0]; self.fb = [0, 0]
        for s in range(2):
            self.fa[s] = n; n += TM * 4
            self.fb[s] = n; n += TN * 4
        self.sa = n; n += TM * 4
        self.sb = n; n += TN * 4
        self.offa = n; n += TM
        self.offb = n; n += TN
        self.wb = n; n += 1
        self.wbb = n; n += 1
        self.ra = n; n += 1
        self.rb = n; n += 1
        self.sc = self.sh = self.sx = -1
        if affine:                       # A-side affine map: scale / shift of this thread's four k (one float4 each per K tile), their LDS address
            self.sc = n; n += 4
            self.sh = n; n += 4
            self.sx = n; n += 1
        self.total = n


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]" if n > 1 else f"v{base}"


def gen_loop(TM, TN, opt):
    affine = bool(opt.get("affine"))
    R = Regs(TM, TN, affine)
    NM = TM * TN
    BM, BN = 32 * TM, 32 * TN
    A_KG = BM * 64                      # bytes of one 16-deep A sub-tile
    B_KG = BN * 64
    B_BASE = 2 * A_KG
    STAGE = 2 * A_KG + 2 * B_KG
    out = []

    def emit(s):
        out.append(s)

    def mfma(set_, step, i, j):
        a = R.fa[set_] + 4 * i + step
        b = R.fb[set_] + 4 * j + step
        c = R.acc + 4 * (i * TN + j)
        return f"v_mfma_f32_16x16x4_f32 {vr(c)}, v{a}, v{b}, {vr(c)}"

    def reads(set_, stage, kg):
        # interleave A / B so the first MFMA's operands arrive first
        ra = [f"ds_read_b128 {vr(R.fa[set_] + 4 * i)}, v{R.ra} offset:{stage * STAGE + kg * A_KG + i * 1024}" for i in range(TM)]
        rb = [f"ds_read_b128 {vr(R.fb[set_] + 4 * j)}, v{R.rb} offset:{stage * STAGE + B_BASE + kg * B_KG + j * 1024}" for j in range(TN)]
        res = []
        for k in range(max(TM, TN)):
            if k < TM: res.append(ra[k])
            if k < TN: res.append(rb[k])
        return res

    def writes(stage):
        w = [f"ds_write_b128 v{R.wb}, {vr(R.sa + 4 * i)} offset:{stage * STAGE + i * 2048}" for i in range(TM)]
        w += [f"ds_write_b128 v{R.wbb}, {vr(R.sb + 4 * j)} offset:{stage * STAGE + B_BASE + j * 2048}" for j in range(TN)]
        return w

    def loads():
        l = [f"global_load_dwordx4 {vr(R.sa + 4 * i)}, v{R.offa + i}, s[{S_A}:{S_A + 1}]" for i in range(TM)]
        l += [f"global_load_dwordx4 {vr(R.sb + 4 * j)}, v{R.offb + j}, s[{S_B}:{S_B + 1}]" for j in range(TN)]
        return l

    def advance():
        return [f"s_add_u32 s{S_A}, s{S_A}, 128", f"s_addc_u32 s{S_A + 1}, s{S_A + 1}, 0",
                f"s_add_u32 s{S_B}, s{S_B}, 128", f"s_addc_u32 s{S_B + 1}, s{S_B + 1}, 0"]

    def phase(set_, fill):
        k = 0
        for step in range(4):
            for i in range(TM):
                for j in range(TN):
                    for ins in fill.get(("pre", k), []): emit(ins)
                    emit(mfma(set_, step, i, j))
                    for ins in fill.get(k, []): emit(ins)
                    k += 1

    def place(fill, items, start, step):
        k = start
        for it in items:
            fill.setdefault(min(k, 4 * NM - 1), []).append(it)
            k += step
        return k

    def iteration(stage, kind):
        P = 4 * NM
        emit(f"; ---- iteration: stage {stage}, {kind}")
        # phase A
        fa = {}
        k = place(fa, reads(1, stage, 1), opt["r1_start"], opt["r1_step"])
        if affine and kind != "last":    # scale / shift of the four k this thread stages in tile t+1 (sx points at them), then sx -> tile t+2
            place(fa, [f"ds_read_b128 {vr(R.sc)}, v{R.sx}", f"ds_read_b128 {vr(R.sh)}, v{R.sx} offset:4096", f"v_add_u32 v{R.sx}, 128, v{R.sx}"], k, 1)
        emit("s_waitcnt lgkmcnt(0)")
        phase(0, fa)
        # phase B
        fb = {}
        if kind != "last":
            w = writes(stage ^ 1)
            if affine:                   # A' = max(0, A * scale + shift) on the staging registers (v_fma_f32 + v_max_f32: what hipcc emits for the compiler loop), B untouched
                wa, wbw = w[:TM], w[TM:]
                ops = []
                for i in range(TM):
                    regs = [R.sa + 4 * i + c for c in range(4)]
                    ops += [f"v_fma_f32 v{r}, v{r}, v{R.sc + c}, v{R.sh + c}" for c, r in enumerate(regs)]
                    ops += [f"v_max_f32_e32 v{r}, 0, v{r}" for r in regs]
                    if i < len(wbw): ops.append(wbw[i])
                    ops.append(wa[i])
                ops += wbw[TM:]
                per = opt.get("affine_per_gap", 2)
                items = [ops[q:q + per] for q in range(0, len(ops), per)]
                items[0] = ["s_waitcnt vmcnt(0)"] + items[0]
                k = place(fb, items, opt["w_start"], 1)
            else:
                items = [["s_waitcnt vmcnt(0)", w[0]]] + [[x] for x in w[1:]]
                k = place(fb, items, opt["w_start"], opt["w_step"])
            if kind == "full":
                l = loads()
                k = place(fb, [advance() + [l[0]]] + [[x] for x in l[1:]], k, opt["l_step"])
            tail = min(opt["tail"], P - 1)
            bpos = max(P - 1 - tail, k)
            fb.setdefault(bpos, []).append(["s_waitcnt lgkmcnt(0)", "s_barrier"])
            place(fb, [[x] for x in reads(0, stage ^ 1, 0)], bpos + opt["r0_gap"], opt["r0_step"])
        # flatten nested lists
        fbf = {k: [y for x in v for y in (x if isinstance(x, list) else [x])] for k, v in fb.items()}
        emit("s_waitcnt lgkmcnt(0)")
        phase(1, fbf)

    # ---------------- prologue: tile 0 is in LDS stage 0 and visible (C++ side); bases point at tile 0
    emit(f"s_cmp_lt_u32 s{S_CNT}, 2")
    emit("s_cbranch_scc1 10f")
    for x in advance(): emit(x)
    for x in loads(): emit(x)
    emit("10:")
    for x in reads(0, 0, 0): emit(x)
    for r in range(TM * TN * 4): emit(f"v_mov_b32 v{R.acc + r}, 0")
    # ---------------- main loop: pairs of full iterations while >= 4 tiles remain
    emit("20:")
    emit(f"s_cmp_lt_u32 s{S_CNT}, 4")
    emit("s_cbranch_scc1 30f")
    iteration(0, "full")
    iteration(1, "full")
    emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 2")
    emit("s_branch 20b")
    # ---------------- tails: 1, 2 or 3 tiles remain, stage 0 next
    emit("30:")
    emit(f"s_cmp_eq_u32 s{S_CNT}, 1")
    emit("s_cbranch_scc1 41f")
    emit(f"s_cmp_eq_u32 s{S_CNT}, 2")
    emit("s_cbranch_scc1 42f")
    iteration(0, "full")                 # 3 remain
    iteration(1, "nolo")
    emit("41:")
    iteration(0, "last")
    emit("s_branch 50f")
    emit("42:")
    iteration(0, "nolo")
    iteration(1, "last")
    emit("50:")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 15")
    return R, out


DEFAULT_OPT = dict(r1_start=1, r1_step=2, w_start=0, w_step=2, l_step=2, tail=16, r0_gap=0, r0_step=1)


def gen_function(TM, TN, suffix="", opt=None):
    o = dict(DEFAULT_OPT)
    if opt: o.update(opt)
    R, lines = gen_loop(TM, TN, o)
    affine = bool(o.get("affine"))
    name = f"nt_asm_loop_{TM}x{TN}{suffix}"
    vt = {1: "unsigned", 2: "u32x2", 4: "u32x4"}
    text = []
    text.append(f"// wave tile {16 * TM} x {16 * TN} (workgroup {32 * TM} x {32 * TN}), {R.total} VGPRs, LDS {2 * (32 * TM + 32 * TN) * 128} B; schedule {o}")
    text.append(f"__device__ __forceinline__ void {name}(f32x4 (&acc)[{TM}][{TN}], const float* pa, const float* pb, int ntiles,")
    text.append(f"        {vt[TM]} offa, {vt[TN]} offb, unsigned wbase_a, unsigned wbase_b, unsigned rbase_a, unsigned rbase_b" + (", unsigned sx" if affine else "") + ") {")
    text.append("    asm volatile(")
    for l in lines:
        text.append(f'        "{l}\\n"')
    outs = []
    for i in range(TM):
        for j in range(TN):
            c = R.acc + 4 * (i * TN + j)
            outs.append(f'"={{v[{c}:{c + 3}]}}"(acc[{i}][{j}])')
    ios = [f'"+{{s[{S_A}:{S_A + 1}]}}"(pa)', f'"+{{s[{S_B}:{S_B + 1}]}}"(pb)', f'"+{{s{S_CNT}}}"(ntiles)']
    if affine: ios.append(f'"+{{v{R.sx}}}"(sx)')
    text.append("        : " + ", ".join(outs) + ",")
    text.append("          " + ", ".join(ios))
    ins = [f'"{{{vr(R.offa, TM)}}}"(offa)', f'"{{{vr(R.offb, TN)}}}"(offb)', f'"{{v{R.wb}}}"(wbase_a)', f'"{{v{R.wbb}}}"(wbase_b)', f'"{{v{R.ra}}}"(rbase_a)', f'"{{v{R.rb}}}"(rbase_b)']
    text.append("        : " + ", ".join(ins))
    clob = [f'"v{r}"' for r in range(R.fa[0], R.offa)] + ([f'"v{r}"' for r in range(R.sc, R.sx)] if affine else [])
    text.append("        : " + ", ".join(clob) + ', "scc", "memory");')
    text.append("}")
    return "\n".join(text)


HEADER = """// gemm_nt_asm_loop.h -- GENERATED by gen_nt_asm.py (do not edit; edit the generator).  The hand-scheduled K loop of the NT fp32 GEMM:
// every instruction of the steady state placed by hand in a v_mfma_f32_16x16x4_f32 gap, physical registers, counted s_waitcnt.
#pragma once
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
"""


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    parts = [HEADER, gen_function(4, 4), "",
             gen_function(4, 2, opt=dict(w_step=1, l_step=1)), "",
             gen_function(2, 2, opt=dict(r1_step=1, w_step=1, l_step=1, tail=6)), "",
             gen_function(4, 4, "_affine", opt=dict(affine=1, l_step=1)), "",
             gen_function(4, 2, "_affine", opt=dict(affine=1, l_step=1, affine_per_gap=3)), ""]
    with open(os.path.join(here, "gemm_nt_asm_loop.h"), "w") as f:
        f.write("\n".join(parts))


if __name__ == "__main__":
    main()
