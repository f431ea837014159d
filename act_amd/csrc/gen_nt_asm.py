#!/usr/bin/env python3
"""Generator of the hand-scheduled GEMM main loops (gfx950 assembly inside one inline-asm statement each) -> gemm_nt_asm_loop.h

Why a generator: a loop is ~1,500 instructions per tile shape whose ORDER is the whole point (which MFMA gap a ds_read / ds_write /
global_load / s_waitcnt / s_barrier sits in); the schedule is described here once and emitted for every tile shape, operand layout and loop phase.

The loop (per wave; WM x WN waves per workgroup, wave tile 16 TM x 16 TN, K tile 32 = two 16-deep sub-tiles `kg`):

    registers   acc[TM][TN] (4 each) | fragment set 0 (A, B) | set 1 | staging (PGR sets of NA + NB float4) | global offsets | LDS bases
    LDS         stage s (0 / 1): A kg0 | A kg1 | B kg0 | B kg1
                K-contiguous operand   [rows][16 k] with the XOR chunk swizzle of gemm_nt16_kernel.h; one ds_read_b128 = four k-steps of one 16-row block
                row-contiguous operand [16 k][rows] as in memory (gemm_q16_kernel.h);                 one ds_read_b128 = one k-step of four 16-row blocks
    iteration t phase A   4 NM MFMAs on set 0 (tile t, kg0); in their gaps: ds_read set 1 <- stage s, kg1
                phase B   4 NM MFMAs on set 1 (tile t, kg1); in their gaps: s_waitcnt vmcnt, ds_write staging -> stage s^1 (tile t+1),
                          global_load tile t+1+PGR -> staging, lgkmcnt(0) + s_barrier TAIL MFMAs before the end, ds_read set 0 <- stage s^1, kg0
    one barrier per 8 NM MFMAs, every memory instruction in an MFMA gap, no s_waitcnt that can stall in steady state.
    PGR = 2 keeps two K tiles in flight in registers (operands streamed from HBM: one iteration does not cover the latency under load).

Products and their order per accumulator are those of the 16-deep compiler loops (sgemm_nt16_kernel / sgemm_q16_kernel): results are bit-identical.
Run: python gen_nt_asm.py   (writes gemm_nt_asm_loop.h next to this file; the header is committed, this script documents it)
"""
import os

S_A, S_B, S_CNT, S_STEP_A, S_STEP_B = 60, 62, 64, 65, 66     # pinned SGPRs: A base (pair), B base (pair), K-tile count, bytes per K tile of A / B
S_NA, S_NB, S_HAS = 68, 70, 72                               # continuous variants: bases of the NEXT output tile's K-tile 0, 1 = there is one


class Cfg:
    def __init__(self, TM, TN, WM=2, WN=2, a_row=False, b_row=False, pgr=1, affine=False, a_sum=False, cont=False, **sched):
        self.TM, self.TN, self.WM, self.WN = TM, TN, WM, WN
        self.a_row, self.b_row, self.pgr, self.affine, self.a_sum = a_row, b_row, pgr, affine, a_sum
        assert not (affine and a_sum)
        self.cont = cont
        assert not cont or pgr == 1
        assert not a_row or TM == 4, "row-contiguous A: 64-row wave extent"
        assert not b_row or TN == 4, "row-contiguous B: 64-column wave extent"
        assert not affine or not a_row
        self.BM, self.BN = WM * 16 * TM, WN * 16 * TN
        self.NA, self.NB = self.BM // 32, self.BN // 32       # float4 per thread and K tile
        self.sched = dict(r1_start=1, r1_step=2, w_start=0, w_step=2, l_step=2, tail=16, r0_gap=0, r0_step=1, affine_per_gap=2)
        self.sched.update(sched)
        n = 0
        self.acc = n; n += TM * TN * 4
        self.fa, self.fb = [0, 0], [0, 0]
        for s in range(2):
            self.fa[s] = n; n += 16 if a_row else TM * 4
            self.fb[s] = n; n += 16 if b_row else TN * 4
        self.sa, self.sb = [], []
        for s in range(pgr):
            self.sa.append(n); n += self.NA * 4
            self.sb.append(n); n += self.NB * 4
        self.clob_end = n
        self.offa = n; n += self.NA
        self.offb = n; n += self.NB
        self.wb = n; n += 1
        self.wbb = n; n += 1
        self.ra = n; n += 1
        self.rb = n; n += 1
        self.sc = self.sh = self.sx = -1
        if affine:                       # A-side affine map: scale / shift of this thread's four k (one float4 each per K tile), their LDS address
            n += n & 1
            self.sc = n; n += 4
            self.sh = n; n += 4
            self.sx = n; n += 1
        self.sv8 = self.sv4 = -1
        if cont:                         # scalars arrive in VGPRs and are moved to the pinned SGPRs inside the asm (see gen_function)
            n += n & 1
            self.sv8 = n; n += 8         # pa.lo, pa.hi, pb.lo, pb.hi, pa_next.lo, pa_next.hi, pb_next.lo, pb_next.hi
            self.sv4 = n; n += 4         # ntiles, step_a, step_b, has_next
        self.bs = -1
        if a_sum:                        # running sums of the A values this thread stages (column sums of A = bias gradients of a weight-gradient GEMM)
            n += n & 1
            self.bs = n; n += 4
        self.total = n
        self.A_KG, self.B_KG = self.BM * 64, self.BN * 64
        self.B_BASE = 2 * self.A_KG
        self.STAGE = 2 * self.A_KG + 2 * self.B_KG

    def name(self):
        lay = {(False, False): "nt", (False, True): "nn", (True, True): "tn"}[(self.a_row, self.b_row)]
        s = f"{lay}_asm_loop_{self.TM}x{self.TN}"
        if (self.WM, self.WN) != (2, 2): s += f"_w{self.WM}x{self.WN}"
        if self.pgr != 1: s += f"_pg{self.pgr}"
        if self.affine: s += "_affine"
        if self.a_sum: s += "_asum"
        if self.cont: s += "_cont"
        return s


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]" if n > 1 else f"v{base}"


def gen_loop(c):
    TM, TN, NM, o = c.TM, c.TN, c.TM * c.TN, c.sched
    P = 4 * NM
    NL = c.NA + c.NB
    out = []
    emit = out.append

    def mfma(set_, step, i, j):
        a = c.fa[set_] + (4 * step + i if c.a_row else 4 * i + step)
        b = c.fb[set_] + (4 * step + j if c.b_row else 4 * j + step)
        acc = c.acc + 4 * (i * TN + j)
        return f"v_mfma_f32_16x16x4_f32 {vr(acc)}, v{a}, v{b}, {vr(acc)}"

    def reads(set_, stage, kg):
        if c.a_row: ra = [f"ds_read_b128 {vr(c.fa[set_] + 4 * s)}, v{c.ra} offset:{stage * c.STAGE + kg * c.A_KG + s * c.BM * 4}" for s in range(4)]
        else:       ra = [f"ds_read_b128 {vr(c.fa[set_] + 4 * i)}, v{c.ra} offset:{stage * c.STAGE + kg * c.A_KG + i * 1024}" for i in range(TM)]
        if c.b_row: rb = [f"ds_read_b128 {vr(c.fb[set_] + 4 * s)}, v{c.rb} offset:{stage * c.STAGE + c.B_BASE + kg * c.B_KG + s * c.BN * 4}" for s in range(4)]
        else:       rb = [f"ds_read_b128 {vr(c.fb[set_] + 4 * j)}, v{c.rb} offset:{stage * c.STAGE + c.B_BASE + kg * c.B_KG + j * 1024}" for j in range(TN)]
        res = []                                              # interleave A / B so the first MFMA's operands arrive first
        for k in range(max(len(ra), len(rb))):
            if k < len(ra): res.append(ra[k])
            if k < len(rb): res.append(rb[k])
        return res

    def writes(stage, sset):
        wa = [f"ds_write_b128 v{c.wb}, {vr(c.sa[sset] + 4 * i)} offset:{stage * c.STAGE + i * (4096 if c.a_row else 2048)}" for i in range(c.NA)]
        wb = [f"ds_write_b128 v{c.wbb}, {vr(c.sb[sset] + 4 * j)} offset:{stage * c.STAGE + c.B_BASE + j * (4096 if c.b_row else 2048)}" for j in range(c.NB)]
        return wa, wb

    def loads(sset):
        l = [f"global_load_dwordx4 {vr(c.sa[sset] + 4 * i)}, v{c.offa + i}, s[{S_A}:{S_A + 1}]" for i in range(c.NA)]
        l += [f"global_load_dwordx4 {vr(c.sb[sset] + 4 * j)}, v{c.offb + j}, s[{S_B}:{S_B + 1}]" for j in range(c.NB)]
        return l

    def advance():
        return [f"s_add_u32 s{S_A}, s{S_A}, s{S_STEP_A}", f"s_addc_u32 s{S_A + 1}, s{S_A + 1}, 0",
                f"s_add_u32 s{S_B}, s{S_B}, s{S_STEP_B}", f"s_addc_u32 s{S_B + 1}, s{S_B + 1}, 0"]

    def phase(set_, fill):
        k = 0
        for step in range(4):
            for i in range(TM):
                for j in range(TN):
                    emit(mfma(set_, step, i, j))
                    for ins in fill.get(k, []): emit(ins)
                    k += 1

    def place(fill, items, start, step):
        k = start
        for it in items:
            fill.setdefault(min(k, P - 1), []).extend(it if isinstance(it, list) else [it])
            k += step
        return k

    def switch_tile():
        return [f"s_mov_b32 s{S_A}, s{S_NA}", f"s_mov_b32 s{S_A + 1}, s{S_NA + 1}", f"s_mov_b32 s{S_B}, s{S_NB}", f"s_mov_b32 s{S_B + 1}, s{S_NB + 1}"]

    def iteration(stage, do_write, do_load, vmwait, reads0=True):
        """stage: LDS stage of tile t.  do_write: tile t+1 exists (staged registers -> stage^1, barrier, set-0 reads); do_load: tile t+1+PGR exists;
        vmwait: loads that may stay in flight when the staged tile is needed"""
        emit(f"; ---- iteration: stage {stage}, write {int(do_write)}, load {do_load}")
        sset = (stage ^ 1) % c.pgr                            # staging set of tile t+1 (and of tile t+1+PGR): tile index mod PGR
        fa = {}
        k = place(fa, reads(1, stage, 1), o["r1_start"], o["r1_step"])
        if c.affine and do_write:        # scale / shift of the four k this thread stages in tile t+1 (sx points at them), then sx -> tile t+2
            place(fa, [f"ds_read_b128 {vr(c.sc)}, v{c.sx}", f"ds_read_b128 {vr(c.sh)}, v{c.sx} offset:4096", f"v_add_u32 v{c.sx}, 128, v{c.sx}"], k, 1)
        emit("s_waitcnt lgkmcnt(0)")
        phase(0, fa)
        fb = {}
        if do_write:
            wa, wbw = writes(stage ^ 1, sset)
            wait = f"s_waitcnt vmcnt({vmwait})"
            if c.affine:                 # A' = max(0, A * scale + shift) on the staging registers (v_fma_f32 + v_max_f32: what hipcc emits for the compiler loop), B untouched
                ops = []
                for i in range(c.NA):
                    regs = [c.sa[sset] + 4 * i + q for q in range(4)]
                    ops += [f"v_fma_f32 v{r}, v{r}, v{c.sc + q}, v{c.sh + q}" for q, r in enumerate(regs)]
                    ops += [f"v_max_f32_e32 v{r}, 0, v{r}" for r in regs]
                    if i < len(wbw): ops.append(wbw[i])
                    ops.append(wa[i])
                ops += wbw[c.NA:]
                per = o["affine_per_gap"]
                items = [ops[q:q + per] for q in range(0, len(ops), per)]
                items[0] = [wait] + items[0]
                k = place(fb, items, o["w_start"], 1)
            elif c.a_sum:                # bs += A chunk, chunks in ascending k order (the order of the compiler loop), each chunk summed before its ds_write
                ops = []
                for i in range(c.NA):
                    ops += [f"v_add_f32_e32 v{c.bs + q}, v{c.bs + q}, v{c.sa[sset] + 4 * i + q}" for q in range(4)]
                    if i < len(wbw): ops.append(wbw[i])
                    ops.append(wa[i])
                ops += wbw[c.NA:]
                items = [ops[q:q + 3] for q in range(0, len(ops), 3)]
                items[0] = [wait] + items[0]
                k = place(fb, items, o["w_start"], 1)
            else:
                w = wa + wbw
                k = place(fb, [[wait, w[0]]] + [[x] for x in w[1:]], o["w_start"], o["w_step"])
            if do_load:                  # "switch": the loads fetch K-tile 0 of the NEXT output tile (continuous variants)
                l = loads(sset)
                k = place(fb, [(switch_tile() if do_load == "switch" else advance()) + [l[0]]] + [[x] for x in l[1:]], k, o["l_step"])
            bpos = max(P - 1 - min(o["tail"], P - 1), k)
            place(fb, [["s_waitcnt lgkmcnt(0)", "s_barrier"]], bpos, 1)
            if reads0:
                place(fb, [[x] for x in reads(0, stage ^ 1, 0)], bpos + o["r0_gap"], o["r0_step"])
        emit("s_waitcnt lgkmcnt(0)")
        phase(1, fb)

    # ---------------- continuous variants: every scalar operand comes in a VGPR (a persistent tile loop keeps wave-uniform values wherever the compiler
    # likes, and a physical-SGPR asm operand fed from a VGPR is a compile error) and is moved to its SGPR here; VALU-writes-SGPR -> VMEM-reads-it needs 5 wait states
    if c.cont:
        emit("s_nop 7")                  # (the compiler's last VALU writes of the operand VGPRs are outside its hazard recogniser's view of this block)
        for q, sreg in enumerate([S_A, S_A + 1, S_B, S_B + 1, S_NA, S_NA + 1, S_NB, S_NB + 1]):
            emit(f"v_readfirstlane_b32 s{sreg}, v{c.sv8 + q}")
        for q, sreg in enumerate([S_CNT, S_STEP_A, S_STEP_B, S_HAS]):
            emit(f"v_readfirstlane_b32 s{sreg}, v{c.sv4 + q}")
        emit("s_nop 7")
    # ---------------- prologue: tile 0 is in LDS stage 0 and visible (C++ side); bases point at tile 0
    for q in range(1, c.pgr + 1):
        emit(f"s_cmp_lt_u32 s{S_CNT}, {q + 1}")
        emit("s_cbranch_scc1 10f")
        for x in advance(): emit(x)
        for x in loads(q % c.pgr): emit(x)
    emit("10:")
    for x in reads(0, 0, 0): emit(x)
    for r in range(TM * TN * 4): emit(f"v_mov_b32 v{c.acc + r}, 0")
    # the kernels raise the wave priority at entry (s_setprio 3): a NEW workgroup must get through its address set-up / K-tile-0 staging while the older
    # workgroup on the CU is in this loop, which never leaves it an issue slot at equal priority (s_memtime stamps: a 3,000-cycle prologue took 97,000 cycles and
    # was still ~5,000 cycles short when the older workgroup finished -- the matrix pipe idles that long at every hand-over).  The loop itself runs at priority 0.
    emit("s_setprio 0")
    FULL = (True, True, NL * (c.pgr - 1))
    if c.pgr == 1:
        # ---------------- main loop: pairs of full iterations while >= 4 tiles remain; tails: 3, 2 or 1 tiles remain, stage 0 next
        emit("20:")
        emit(f"s_cmp_lt_u32 s{S_CNT}, 4")
        emit("s_cbranch_scc1 30f")
        iteration(0, *FULL)
        iteration(1, *FULL)
        emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 2")
        emit("s_branch 20b")
        emit("30:")
        if c.cont:
            # continuous K stream (even K-tile counts only): the last two iterations of this output tile fetch K-tile 0 of the NEXT one and leave it in
            # LDS stage 0 behind a barrier -- the next call starts its MFMAs at once instead of waiting for a global load + LDS store + barrier
            emit(f"s_cmp_eq_u32 s{S_HAS}, 0")
            emit("s_cbranch_scc1 31f")
            iteration(0, True, "switch", 0)
            iteration(1, True, False, 0, reads0=False)
            emit("s_branch 50f")
            emit("31:")
        emit(f"s_cmp_eq_u32 s{S_CNT}, 1")
        emit("s_cbranch_scc1 41f")
        emit(f"s_cmp_eq_u32 s{S_CNT}, 2")
        emit("s_cbranch_scc1 42f")
        iteration(0, *FULL)                       # 3 remain
        iteration(1, True, False, 0)
        emit("41:")
        iteration(0, False, False, 0)
        emit("s_branch 50f")
        emit("42:")
        iteration(0, True, False, 0)
        iteration(1, False, False, 0)
    else:
        # ---------------- PGR 2: a full iteration needs tile t+3; main loop while >= 5 remain; tails: 4, 3, 2 or 1 remain, stage 0 next
        emit("20:")
        emit(f"s_cmp_lt_u32 s{S_CNT}, 5")
        emit("s_cbranch_scc1 30f")
        iteration(0, *FULL)
        iteration(1, *FULL)
        emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 2")
        emit("s_branch 20b")
        emit("30:")
        emit(f"s_cmp_eq_u32 s{S_CNT}, 1")
        emit("s_cbranch_scc1 41f")
        emit(f"s_cmp_eq_u32 s{S_CNT}, 2")
        emit("s_cbranch_scc1 42f")
        emit(f"s_cmp_eq_u32 s{S_CNT}, 3")
        emit("s_cbranch_scc1 43f")
        iteration(0, *FULL)                       # 4 remain: tiles t+1, t+2 in flight, t+3 loaded here
        iteration(1, True, False, NL)
        emit("42:")
        iteration(0, True, False, 0)              # 2 remain: the last tile is the only one in flight
        iteration(1, False, False, 0)
        emit("s_branch 50f")
        emit("43:")
        iteration(0, True, False, NL)             # 3 remain
        iteration(1, True, False, 0)
        emit("41:")
        iteration(0, False, False, 0)
    emit("50:")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
    emit("s_nop 15")
    emit("s_nop 15")
    return out


def gen_function(c):
    lines = gen_loop(c)
    vt = {1: "unsigned", 2: "u32x2", 4: "u32x4", 8: "u32x8"}
    TM, TN = c.TM, c.TN
    text = []
    lay = ("A [K][M]" if c.a_row else "A [M][K]") + ", " + ("B [K][N]" if c.b_row else "B [N][K]")
    text.append(f"// {lay}; {c.WM} x {c.WN} waves, wave tile {16 * TM} x {16 * TN} (workgroup {c.BM} x {c.BN}), {c.total} VGPRs, LDS {2 * c.STAGE} B, "
                f"global prefetch {c.pgr}; schedule {c.sched}")
    if c.cont:
        text.append(f"__device__ __forceinline__ void {c.name()}(f32x4 (&acc)[{TM}][{TN}], u32x8 bases, u32x4 scalars,   // bases: pa, pb, pa_next, pb_next (lo, hi each); scalars: ntiles, step_a, step_b, has_next")
        text.append(f"        {vt[c.NA]} offa, {vt[c.NB]} offb, unsigned wbase_a, unsigned wbase_b, unsigned rbase_a, unsigned rbase_b) {{")
    else:
        text.append(f"__device__ __forceinline__ void {c.name()}(f32x4 (&acc)[{TM}][{TN}], const float* pa, const float* pb, int ntiles, unsigned step_a, unsigned step_b,")
        text.append(f"        {vt[c.NA]} offa, {vt[c.NB]} offb, unsigned wbase_a, unsigned wbase_b, unsigned rbase_a, unsigned rbase_b" + (", unsigned sx" if c.affine else "") + (", f32x4& bsum" if c.a_sum else "") + ") {")
    text.append("    asm volatile(")
    for l in lines:
        text.append(f'        "{l}\\n"')
    outs = []
    for i in range(TM):
        for j in range(TN):
            a = c.acc + 4 * (i * TN + j)
            outs.append(f'"={{v[{a}:{a + 3}]}}"(acc[{i}][{j}])')
    if c.cont:
        text.append("        : " + ", ".join(outs))
        ins = [f'"{{{vr(c.sv8, 8)}}}"(bases)', f'"{{{vr(c.sv4, 4)}}}"(scalars)', f'"{{{vr(c.offa, c.NA)}}}"(offa)', f'"{{{vr(c.offb, c.NB)}}}"(offb)',
               f'"{{v{c.wb}}}"(wbase_a)', f'"{{v{c.wbb}}}"(wbase_b)', f'"{{v{c.ra}}}"(rbase_a)', f'"{{v{c.rb}}}"(rbase_b)']
        text.append("        : " + ", ".join(ins))
    else:
        ios = [f'"+{{s[{S_A}:{S_A + 1}]}}"(pa)', f'"+{{s[{S_B}:{S_B + 1}]}}"(pb)', f'"+{{s{S_CNT}}}"(ntiles)']
        if c.affine: ios.append(f'"+{{v{c.sx}}}"(sx)')
        if c.a_sum: ios.append(f'"+{{v[{c.bs}:{c.bs + 3}]}}"(bsum)')
        text.append("        : " + ", ".join(outs) + ",")
        text.append("          " + ", ".join(ios))
        ins = [f'"{{s{S_STEP_A}}}"(step_a)', f'"{{s{S_STEP_B}}}"(step_b)', f'"{{{vr(c.offa, c.NA)}}}"(offa)', f'"{{{vr(c.offb, c.NB)}}}"(offb)',
               f'"{{v{c.wb}}}"(wbase_a)', f'"{{v{c.wbb}}}"(wbase_b)', f'"{{v{c.ra}}}"(rbase_a)', f'"{{v{c.rb}}}"(rbase_b)']
        text.append("        : " + ", ".join(ins))
    clob = [f'"v{r}"' for r in range(c.fa[0], c.clob_end)] + ([f'"v{r}"' for r in range(c.sc, c.sx)] if c.affine else [])
    if c.cont: clob += [f'"s{r}"' for r in range(S_A, S_HAS + 1)]
    text.append("        : " + ", ".join(clob) + ', "scc", "memory");')
    text.append("}")
    return "\n".join(text)


HEADER = """// gemm_nt_asm_loop.h -- GENERATED by gen_nt_asm.py (do not edit; edit the generator).  The hand-scheduled K loops of the fp32 GEMMs:
// every instruction of the steady state placed by hand in a v_mfma_f32_16x16x4_f32 gap, physical registers, counted s_waitcnt.
#pragma once
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
"""

SMALL = dict(w_step=1, l_step=1)
VARIANTS = [
    Cfg(4, 4), Cfg(4, 2, **SMALL), Cfg(2, 2, r1_step=1, tail=6, **SMALL),                      # NT 128x128, 128x64, 64x64
    Cfg(4, 4, affine=True, l_step=1), Cfg(4, 2, affine=True, l_step=1, affine_per_gap=3),      # NT + A-side affine on load
    Cfg(4, 4, pgr=2), Cfg(4, 2, pgr=2, **SMALL),                                               # NT, two K tiles in flight
    Cfg(4, 4, cont=True), Cfg(4, 2, cont=True, **SMALL), Cfg(2, 2, cont=True, r1_step=1, tail=6, **SMALL),   # NT, continuous K stream across output tiles (persistent workgroups)
    Cfg(4, 4, b_row=True), Cfg(2, 4, b_row=True, **SMALL),                                     # NN 128x128, 64x128 (2 x 2 waves)
    Cfg(2, 4, WM=4, WN=1, b_row=True, **SMALL), Cfg(1, 4, WM=4, WN=1, b_row=True, r1_step=1, tail=6, **SMALL),   # NN 128x64, 64x64 (4 x 1 waves)
    Cfg(4, 4, a_row=True, b_row=True),                                                         # TN 128x128
    Cfg(4, 4, a_row=True, b_row=True, a_sum=True, l_step=1),                                   # TN 128x128 + column sums of A (grouped weight gradients: bias gradient)
]


def render():
    parts = [HEADER]
    for c in VARIANTS:
        parts += [gen_function(c), ""]
    return "\n".join(parts)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "gemm_nt_asm_loop.h"), "w") as f:
        f.write(render())


if __name__ == "__main__":
    main()
