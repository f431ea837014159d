// attention.hip -- fused multi-head self-attention for the short sequences of ACT (S = 14 / 64 / 128 tokens,
// head_dim 64; models/act.py:57-69 == utils/transformer_layers.py:170-182):  softmax(q k^T * hd^-1/2) v.
//
// Forward (MFMA, fp32 exact): the packed projection output qkv [B,S,3,H,hd] is consumed in place and the result is
// written in the [B,S,H*hd] layout the output projection GEMM reads -- no permutes, no S x S matrix in memory.
// One wave owns 32 query rows of one (cloud, head).  It computes the TRANSPOSED score tile S^T = K Q^T with
// v_mfma_f32_32x32x2_f32 so that, in the MFMA C/D layout, every lane ends up holding all scores of ONE query
// (its lane&31) for 16 keys per 32-key tile: softmax is then a pure in-register reduction plus a single
// lane^32 exchange, and the probabilities are already in the B-operand layout of the second product
// O^T = V^T P^T.  K and V are staged once per (cloud, head) in LDS ([key][hd+4]: conflict-free b128 A-operand
// reads for K, conflict-free b32 reads for V); Q rows are loaded straight into registers.
//
// Backward (student only: S = 14 and 64): one workgroup per (cloud, head), everything resident in LDS,
// P recomputed from the saved log-sum-exp, register-blocked 4x4 VALU micro-tiles; writes dqkv in the packed layout.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef ACT_ATTN_SOFTMAX_LEAN
#define ACT_ATTN_SOFTMAX_LEAN 1          // 0: the round-1..5 form of the forward's online softmax (A/B builds: ACT_HIPCC_EXTRA=-DACT_ATTN_SOFTMAX_LEAN=0)
#endif

// General operand description: queries come from `q` (Sq rows per cloud), keys/values from up to two row segments
// (segment 0: S0 rows, e.g. the prompt tokens of the teacher; segment 1: S1 rows).  Packed qkv is the special case
// S0 = 0, k1 = qkv + H*hd, v1 = qkv + 2*H*hd.  All pointers address head 0; head h adds h*HD.
struct AttnFwdArgs {
    const float* q; const float* k0; const float* v0; const float* k1; const float* v1;
    long long q_bs, kv0_bs, kv1_bs;      // per-cloud strides (floats)
    int ldq, ld0, ld1;                   // row strides (floats)
    int B, H, Sq, S0, S1;
    float scale;
    float* out; float* lse;              // out [B, Sq, H*HD]; lse [B, H, Sq] or null
    unsigned short* out_hi; unsigned short* out_lo;   // optional: out ALSO / ONLY (out == null) as (hi, lo) bf16 planes of the same layout (opt-in split-bf16 teacher)
    int slot_prio;                       // > 0: static wave priority from the workgroup's CU slot (attn_slot_prio)
    int stagger, stagger_mod;            // start-up de-phasing of the workgroups that share a CU: sleep (slot % stagger_mod) * stagger * 1024 cycles (0 = off)
};

// JT = 32-key tiles per LDS-resident key chunk (Sk <= 128: one chunk; longer sequences: chunks of 128 keys with an online
// softmax -- running max m and sum l per query; the rescale of the output accumulators is a per-lane scalar because every
// accumulator register of a lane belongs to the same query).  QT = 32-query tiles per (cloud, head) handled by a workgroup.
// VT (round 5, opt-in, measured slower -- see g_attn_vt): V is staged TRANSPOSED, Vt[d][key] with a row pitch of ROWS + 4 floats, so that the A operand of the second product -- V^T[d = lane][key(r)]
// for the four consecutive keys of MFMA steps r = 4g .. 4g+3 -- is ONE ds_read_b128 per four MFMAs, exactly like the K fragments of the first product.
// With V as [key][d] every MFMA step needed its own ds_read_b32, and the compiler emitted `ds_read2_b32; s_waitcnt lgkmcnt(0); 2 x v_mfma` sixteen times
// per chunk: an exposed LDS round trip per 128 MFMA cycles.  The transpose costs nothing: a lane = one head-dimension column loads the four keys of a
// quad with four coalesced dword loads (a wave reads a 256-byte row segment per instruction) and writes them with one conflict-free ds_write_b128.
// Same products in the same order: bit-identical to VT = false (tests/test_gpu_dense.py).
// NW (round 6) = waves per workgroup: 4, or 2 = ONE pair per workgroup at QT = 2 (Sq = 64: only the two query tiles of a pair share K / V, so a barrier need not couple two
// pairs; 6 independent workgroups per CU instead of 3).  PRIO: s_setprio 1 around the MFMA bursts (waves in their softmax / staging phase yield the issue port).
// The workgroups of a single-round launch that share a CU start together and run their staging / MFMA / softmax phases in lock step (one resident round of
// waves: the matrix pipe idles while all of them are in softmax).  De-phasing: the workgroup in CU slot t (HW_ID.TG_ID, wave-uniform, the same for every wave
// of the workgroup; stagger < 0: slot taken from blockIdx.x / 256 instead) sleeps (t % mod) * |stagger| * 1024 cycles before its first load.
__device__ __forceinline__ void attn_stagger(int stagger, int mod) {
    if (stagger == 0) return;
    unsigned slot;
    if (stagger > 0) { unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); slot = (hwid >> 16) & 15u; }
    else { slot = blockIdx.x >> 8; stagger = -stagger; }
    const int n = (int)(slot % (unsigned)mod) * stagger;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
}
// DIAG (dev ablation, ACT_ATTN_FWD_DIAG, results are WRONG when non-zero): 1 = every pair reads pair 0's operands (all fetches hit L2), 2 = no output store,
// 4 = no exponentials (p = score), 8 = second product skipped, 16 = first product skipped, 32 = no barriers after the first chunk
// Static priority by CU slot: the waves of the workgroups that share a SIMD run the same phases (MFMA burst, softmax, MFMA burst, staging) and the arbiter
// serves equal priorities in turn, so they stay in lock step and the matrix pipe idles while all of them are in their VALU phase (ablation: time = skeleton
// + MFMA, no overlap).  With DIFFERENT priorities the highest one owns the matrix pipe whenever it wants it and the others fill its VALU phases.
__device__ __forceinline__ void attn_slot_prio(int mode) {
    if (mode <= 0) return;
    unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const unsigned slot = mode == 2 ? (hwid & 15u) : ((hwid >> 16) & 15u);       // 1: workgroup slot on the CU (TG_ID); 2: wave slot on the SIMD (WAVE_ID)
    switch (slot % 3u) {
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(1); break;
    }
}
template <int HD, int JT, int QT, bool VT, int NW = 4, bool PRIO = false, int DIAG = 0>
__global__ __launch_bounds__(NW * 64, (JT == 1 ? 3 : 2)) void attn_fwd_kernel(const AttnFwdArgs a) {
    constexpr int NTHR = NW * 64;
    constexpr int LDK = HD + 4;
    constexpr int PAIRS = NW / QT;                                  // (cloud, head) pairs per workgroup (NW = 4: 4 / 2 / 1 / 1 for QT = 1 .. 4)
    static_assert(PAIRS >= 1, "a workgroup holds at least the query tiles of one pair");
    constexpr int ROWS = JT * 32;
    constexpr int LDT = ROWS + 4;                                   // Vt row pitch: 4 (mod 32) dwords -> conflict-free b128 reads / writes
    constexpr int KSZ = ROWS * LDK, VSZ = VT ? HD * LDT : ROWS * LDK, PSZ = KSZ + VSZ;     // floats per pair: K image, V image
    static_assert((ROWS * (HD / 4)) % NTHR == 0, "staging: whole iterations per pair");
    extern __shared__ __attribute__((aligned(16))) float smem[];    // [PAIRS][K image | V image]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, Sk = a.S0 + a.S1;
    attn_stagger(a.stagger, a.stagger_mod);
    attn_slot_prio(a.slot_prio);
    const long long npairs = (long long)a.B * H;
    const long long pair0 = (long long)blockIdx.x * PAIRS;
    const int qblock = blockIdx.y * (QT * 32);                       // first query row of this workgroup

    const int pl = wave / QT, qt = wave % QT;
    const long long pr = pair0 + pl;
    const bool active = pl < PAIRS && pr < npairs && qblock + qt * 32 < a.Sq;
    const int b_true = active ? (int)(pr / H) : 0, h_true = active ? (int)(pr % H) : 0;
    const int b = (DIAG & 1) ? 0 : b_true, h = (DIAG & 1) ? 0 : h_true;
    const float* Ks = smem + (size_t)pl * PSZ;
    const float* Vs = smem + (size_t)pl * PSZ + KSZ;
    const int ql = lane & 31, half = lane >> 5;
    const int q = qblock + qt * 32 + ql;
    const float scale = a.scale;
    const float sc2 = scale * 1.44269504088896340736f;             // scale * log2(e)

    // ---- Q operand: lane (q, half) holds Q[q][half*HD/2 + s], s = 0..HD/2-1
    float qreg[HD / 2];
    if (active) {
        const float* qp = a.q + (size_t)b * a.q_bs + (size_t)min(q, a.Sq - 1) * a.ldq + h * HD + half * (HD / 2);
#pragma unroll
        for (int s4 = 0; s4 < HD / 8; ++s4) {
            float4 t = *reinterpret_cast<const float4*>(qp + s4 * 4);
            if (q >= a.Sq) t = make_float4(0.f, 0.f, 0.f, 0.f);
            qreg[s4 * 4 + 0] = t.x; qreg[s4 * 4 + 1] = t.y; qreg[s4 * 4 + 2] = t.z; qreg[s4 * 4 + 3] = t.w;
        }
    }
    f32x16 o[HD / 32];
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = -3.0e38f, l = 0.f;

    // Staging is software-pipelined across key chunks (round 4): the float4 of chunk c+1 are requested right after chunk c went to LDS and stay in flight in
    // registers while chunk c is multiplied -- with one 32-key chunk resident per step (JT = 1: the teacher's 64 x 128 prefix shape, 4 chunks) the loop was
    // load-latency bound: every chunk paid a full L2 / fabric round trip between its two barriers (MfmaUtil 36 %).  ROWS * HD / 4 is a multiple of 256, so all
    // threads of an iteration work on the same pair: the (cloud, head) split is wave-uniform 32-bit arithmetic done once per pair, not a 64-bit division per float4.
    // Only for JT = 1 (8 float4 per thread at QT >= 2; the larger resident chunks would need 64-128 staging registers and lose a wave per SIMD or spill).
    constexpr int ITS = ROWS * (HD / 4) / NTHR;
    constexpr bool PF = (JT <= 2 && QT >= 2);
    float4 kreg[PAIRS][ITS], vreg[PAIRS][ITS];
    // VT: staging item idx = (key quad idx / HD, column idx % HD): the four keys kc + 4*quad + 0..3 of one column, zero beyond Sk.  The launcher takes
    // this path only when S0 % ROWS == 0, so a chunk lies in ONE key segment: base pointer and row pitch are scalars, an address is one 32-bit offset;
    // keys beyond Sk re-read the chunk's first row (always valid) and are zeroed afterwards -- no divergent control flow around the loads.
    auto load_vt = [&](const float* v0p, const float* v1p, bool live, int kc, int idx) -> float4 {
        const bool seg0 = kc < a.S0;
        const float* vb = seg0 ? v0p : v1p;
        const unsigned ld = seg0 ? a.ld0 : a.ld1;
        const int r0 = seg0 ? kc : kc - a.S0;
        const int d = idx % HD, q4 = 4 * (idx / HD);
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = kc + q4 + j < Sk;
            t[j] = live ? vb[(unsigned)(r0 + (ok ? q4 + j : 0)) * ld + d] : 0.f;
        }
        return make_float4(t[0], t[1], t[2], t[3]);
    };
    // ... zeroed when the quad goes to LDS, not when it is requested (a select on the loaded value would put `s_waitcnt vmcnt(0)` right behind the
    // prefetch of the NEXT chunk and serialise it with this chunk's products)
    auto mask_vt = [&](float4 v, int kc, int idx) -> float4 {
        const int k0 = kc + 4 * (idx / HD);
        v.x = k0 + 0 < Sk ? v.x : 0.f; v.y = k0 + 1 < Sk ? v.y : 0.f; v.z = k0 + 2 < Sk ? v.z : 0.f; v.w = k0 + 3 < Sk ? v.w : 0.f;
        return v;
    };
    auto v_lds_offset = [&](int idx) -> int {                        // where staging item idx of a pair goes inside the pair's V image
        return VT ? (idx % HD) * LDT + 4 * (idx / HD) : (idx / (HD / 4)) * LDK + (idx % (HD / 4)) * 4;
    };
    // (cloud, head) of every pair of the workgroup: wave-uniform, divided ONCE (a 32-bit division is ~25 VALU instructions, and it sat inside the chunk loop)
    int pb[PAIRS], ph[PAIRS];
#pragma unroll
    for (int p2 = 0; p2 < PAIRS; ++p2) {
        const unsigned pr2 = (unsigned)pair0 + p2;
        const bool live = (long long)pr2 < npairs;
        const unsigned b2 = live ? pr2 / (unsigned)H : 0u;
        pb[p2] = (DIAG & 1) ? 0 : __builtin_amdgcn_readfirstlane((int)b2);
        ph[p2] = (DIAG & 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(live ? pr2 - b2 * (unsigned)H : 0u));
    }
    const bool all_live = pair0 + PAIRS <= npairs;
    auto load_chunk = [&](int kc) {
        // fast path (round 6): a FULL chunk inside ONE key segment, every pair of the workgroup live -- all rows valid, base pointer and row pitch are
        // wave-uniform: an SGPR base plus one 32-bit lane offset per item instead of per-item compares, 64-bit address arithmetic and exec-masked
        // branches (the PMC pass showed 6.5 VALU instructions per MFMA at the teacher shape, 110 of the 290 per chunk in this staging code)
        const bool in0 = kc + ROWS <= a.S0, in1 = kc >= a.S0 && kc + ROWS <= Sk;
        if (!VT && all_live && (in0 || in1)) {
            const unsigned ld = in0 ? a.ld0 : a.ld1;
            const int r0 = in0 ? kc : kc - a.S0;
#pragma unroll
            for (int p2 = 0; p2 < PAIRS; ++p2) {
                const size_t po = (in0 ? (size_t)pb[p2] * a.kv0_bs : (size_t)pb[p2] * a.kv1_bs) + (size_t)ph[p2] * HD + (size_t)r0 * ld;
                const float* kb = (in0 ? a.k0 : a.k1) + po; const float* vb = (in0 ? a.v0 : a.v1) + po;
#pragma unroll
                for (int it = 0; it < ITS; ++it) {
                    const int idx = tid + NTHR * it;
                    const unsigned of = (unsigned)(idx / (HD / 4)) * ld + (idx % (HD / 4)) * 4;
                    kreg[p2][it] = *reinterpret_cast<const float4*>(kb + of);
                    vreg[p2][it] = *reinterpret_cast<const float4*>(vb + of);
                }
            }
            return;
        }
#pragma unroll
        for (int p2 = 0; p2 < PAIRS; ++p2) {
            const unsigned pr2 = (unsigned)pair0 + p2;
            const bool live = (long long)pr2 < npairs;
            const unsigned b2 = (unsigned)pb[p2], h2 = (unsigned)ph[p2];
            const float* k0p = a.k0 + (size_t)b2 * a.kv0_bs + h2 * HD; const float* v0p = a.v0 + (size_t)b2 * a.kv0_bs + h2 * HD;
            const float* k1p = a.k1 + (size_t)b2 * a.kv1_bs + h2 * HD; const float* v1p = a.v1 + (size_t)b2 * a.kv1_bs + h2 * HD;
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const int idx = tid + NTHR * it;
                const int c4 = idx % (HD / 4), rl = idx / (HD / 4), row = kc + rl;
                float4 kx = make_float4(0.f, 0.f, 0.f, 0.f), vx = kx;
                if (live && row < Sk) {
                    if (row < a.S0) {
                        const unsigned of = (unsigned)row * a.ld0 + c4 * 4;
                        kx = *reinterpret_cast<const float4*>(k0p + of); if (!VT) vx = *reinterpret_cast<const float4*>(v0p + of);
                    } else {
                        const unsigned of = (unsigned)(row - a.S0) * a.ld1 + c4 * 4;
                        kx = *reinterpret_cast<const float4*>(k1p + of); if (!VT) vx = *reinterpret_cast<const float4*>(v1p + of);
                    }
                }
                if (VT) vx = load_vt(v0p, v1p, live, kc, idx);
                kreg[p2][it] = kx; vreg[p2][it] = vx;
            }
        }
    };
    if constexpr (PF) load_chunk(0);
    for (int kc = 0; kc < Sk; kc += ROWS) {
        if (kc > 0 && !(DIAG & 32)) __syncthreads();                 // previous chunk fully consumed
        if constexpr (PF) {
            // ---- this chunk of K and V (zero rows beyond Sk) for every pair of the workgroup: registers -> LDS, then request the next chunk
#pragma unroll
            for (int p2 = 0; p2 < PAIRS; ++p2) {
                float* kd = smem + (size_t)p2 * PSZ; float* vd = kd + KSZ;
#pragma unroll
                for (int it = 0; it < ITS; ++it) {
                    const int idx = tid + NTHR * it;
                    const int c4 = idx % (HD / 4), rl = idx / (HD / 4);
                    *reinterpret_cast<float4*>(kd + rl * LDK + c4 * 4) = kreg[p2][it];
                    *reinterpret_cast<float4*>(vd + v_lds_offset(idx)) = VT ? mask_vt(vreg[p2][it], kc, idx) : vreg[p2][it];
                }
            }
            if (kc + ROWS < Sk) load_chunk(kc + ROWS);
        } else {
            // ---- stage this chunk of K and V for every pair of the workgroup (zero rows beyond Sk).  ROWS * HD / 4 is a multiple of 256, so all threads
            // of an iteration work on the same pair: the (cloud, head) split is wave-uniform 32-bit arithmetic done once per pair, not a 64-bit division
            // per staged float4
    #pragma unroll
            for (int p2 = 0; p2 < PAIRS; ++p2) {
                const unsigned pr2 = (unsigned)pair0 + p2;
                const bool live = (long long)pr2 < npairs;
                const unsigned b2 = live ? pr2 / (unsigned)H : 0u, h2 = live ? pr2 - b2 * (unsigned)H : 0u;
                const float* k0p = a.k0 + (size_t)b2 * a.kv0_bs + h2 * HD; const float* v0p = a.v0 + (size_t)b2 * a.kv0_bs + h2 * HD;
                const float* k1p = a.k1 + (size_t)b2 * a.kv1_bs + h2 * HD; const float* v1p = a.v1 + (size_t)b2 * a.kv1_bs + h2 * HD;
                float* kd = smem + (size_t)p2 * PSZ; float* vd = kd + KSZ;
    #pragma unroll
                for (int it = 0; it < ROWS * (HD / 4) / NTHR; ++it) {
                    const int idx = tid + NTHR * it;
                    const int c4 = idx % (HD / 4), rl = idx / (HD / 4), row = kc + rl;
                    float4 kx = make_float4(0.f, 0.f, 0.f, 0.f), vx = kx;
                    if (live && row < Sk) {
                        if (row < a.S0) {
                            const unsigned of = (unsigned)row * a.ld0 + c4 * 4;
                            kx = *reinterpret_cast<const float4*>(k0p + of); if (!VT) vx = *reinterpret_cast<const float4*>(v0p + of);
                        } else {
                            const unsigned of = (unsigned)(row - a.S0) * a.ld1 + c4 * 4;
                            kx = *reinterpret_cast<const float4*>(k1p + of); if (!VT) vx = *reinterpret_cast<const float4*>(v1p + of);
                        }
                    }
                    if (VT) vx = mask_vt(load_vt(v0p, v1p, live, kc, idx), kc, idx);
                    *reinterpret_cast<float4*>(kd + rl * LDK + c4 * 4) = kx;
                    *reinterpret_cast<float4*>(vd + v_lds_offset(idx)) = vx;
                }
            }
        }
        if (!(DIAG & 32) || kc == 0) __syncthreads();
        if (!active) continue;                                       // idle waves only help staging
        // ---- S^T tiles: acc[jt][r] = score(key = kc + jt*32 + (r&3) + 8*(r>>2) + 4*half, query = ql)
        f32x16 acc[JT];
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jt][r] = 0.f;
            const float* kp = Ks + (size_t)(jt * 32 + ql) * LDK + half * (HD / 2);
#pragma unroll
            for (int s4 = 0; s4 < HD / 8; ++s4) {
                const float4 kk = *reinterpret_cast<const float4*>(kp + s4 * 4);
                if constexpr ((DIAG & 16) != 0) { acc[jt][s4] += kk.x * qreg[s4 * 4] + kk.y + kk.z + kk.w; continue; }
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.x, qreg[s4 * 4 + 0], acc[jt], 0, 0, 0);
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.y, qreg[s4 * 4 + 1], acc[jt], 0, 0, 0);
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.z, qreg[s4 * 4 + 2], acc[jt], 0, 0, 0);
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.w, qreg[s4 * 4 + 3], acc[jt], 0, 0, 0);
            }
        }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        // ---- online softmax over keys (registers + one lane^32 exchange)
        float mc = -3.0e38f;
        if (kc + ROWS <= Sk) {                                       // (wave-uniform) a full chunk has no key to mask: 3 VALU instructions per score less
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mc = fmaxf(mc, acc[jt][r]);
        } else {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kc + jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float v = key < Sk ? acc[jt][r] : -3.0e38f;
                    acc[jt][r] = v;
                    mc = fmaxf(mc, v);
                }
        }
        mc = fmaxf(mc, __shfl_xor(mc, 32));
        const float mn = fmaxf(m, mc);
#if ACT_ATTN_SOFTMAX_LEAN
        // exp(scale (s - m)) = exp2(s c - m c), c = scale log2(e): ONE packed FMA per two scores in front of v_exp_f32 instead of sub + mul + mul per score, and the
        // row sum as packed adds.  On gfx950 no VALU instruction overlaps an f32 MFMA of the same SIMD (profiles/r06_mfma_valu_kinds.txt): every instruction removed
        // here is matrix-pipe time.  (m c is rounded once: a score equal to the maximum gives exp2(+-1 ulp of m c) instead of exactly 1 -- 1e-7.)
        const float mcn = mn * sc2;
        const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m, sc2, -mcn));       // 0 on the first chunk (m = -huge)
        m = mn;
        f32x2 lc2 = f32x2{0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 t = __builtin_elementwise_fma(f32x2{acc[jt][r], acc[jt][r + 1]}, f32x2{sc2, sc2}, f32x2{-mcn, -mcn});
                const f32x2 p = (DIAG & 4) ? t : f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};     // masked keys: exp2(-huge) == 0
                acc[jt][r] = p[0]; acc[jt][r + 1] = p[1];
                lc2 += p;
            }
        float lc = lc2[0] + lc2[1];
#else
        const float alpha = __expf(scale * (m - mn));                // 0 on the first chunk (m = -huge)
        m = mn;
        float lc = 0.f;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = (DIAG & 4) ? scale * (acc[jt][r] - m) : __expf(scale * (acc[jt][r] - m));    // masked keys: exp(-huge) == 0
                acc[jt][r] = p;
                lc += p;
            }
#endif
        lc += __shfl_xor(lc, 32);
        l = l * alpha + lc;
        if (kc > 0) {
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        // ---- O^T += V^T P^T : o[dt][r] = out(d = dt*32 + (r&3) + 8*(r>>2) + 4*half, query = ql)
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
        if constexpr (VT) {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {                                   // steps r = 4g .. 4g+3: keys jt*32 + 8g + 4*half + 0..3
                    float4 vv[HD / 32];
#pragma unroll
                    for (int dt = 0; dt < HD / 32; ++dt)
                        vv[dt] = *reinterpret_cast<const float4*>(Vs + (size_t)(dt * 32 + ql) * LDT + jt * 32 + 8 * g + 4 * half);
#pragma unroll
                    for (int dt = 0; dt < HD / 32; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[dt].x, acc[jt][4 * g + 0], o[dt], 0, 0, 0);
#pragma unroll
                    for (int dt = 0; dt < HD / 32; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[dt].y, acc[jt][4 * g + 1], o[dt], 0, 0, 0);
#pragma unroll
                    for (int dt = 0; dt < HD / 32; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[dt].z, acc[jt][4 * g + 2], o[dt], 0, 0, 0);
#pragma unroll
                    for (int dt = 0; dt < HD / 32; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[dt].w, acc[jt][4 * g + 3], o[dt], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;   // this half-wave's k index for step r
                    const float* vp = Vs + (size_t)key * LDK + ql;
#pragma unroll
                    for (int dt = 0; dt < HD / 32; ++dt) {
                        if constexpr ((DIAG & 8) != 0) { o[dt][r] += vp[dt * 32] * acc[jt][r]; continue; }
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[dt * 32], acc[jt][r], o[dt], 0, 0, 0);
                    }
                }
        }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    if (active && q < a.Sq) {
        const float inv_l = 1.0f / l;
        const size_t obase = ((size_t)b_true * a.Sq + q) * (H * HD) + h_true * HD;
        float* op = a.out ? a.out + obase : nullptr;
        if constexpr ((DIAG & 2) != 0) { if (l != 12345.678f) op = nullptr; }
#pragma unroll
        for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 t;
                t.x = o[dt][g * 4 + 0] * inv_l; t.y = o[dt][g * 4 + 1] * inv_l;
                t.z = o[dt][g * 4 + 2] * inv_l; t.w = o[dt][g * 4 + 3] * inv_l;
                if (op) *reinterpret_cast<float4*>(op + dt * 32 + 8 * g + 4 * half) = t;
                if (a.out_hi) {                                   // same rounding as split_bf16x2_kernel (gemm_bf16x3.hip): bit-identical to splitting `out`
                    const float e[4] = {t.x, t.y, t.z, t.w};
                    unsigned hh[4], ll[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        unsigned w_ = __float_as_uint(e[u]); w_ += 0x7FFFu + ((w_ >> 16) & 1u); hh[u] = w_ >> 16;
                        unsigned x_ = __float_as_uint(e[u] - __uint_as_float(hh[u] << 16)); x_ += 0x7FFFu + ((x_ >> 16) & 1u); ll[u] = x_ >> 16;
                    }
                    const size_t oi = obase + dt * 32 + 8 * g + 4 * half;
                    *reinterpret_cast<uint2*>(a.out_hi + oi) = make_uint2(hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16));
                    *reinterpret_cast<uint2*>(a.out_lo + oi) = make_uint2(ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16));
                }
            }
        if (a.lse && half == 0) a.lse[((size_t)b_true * H + h_true) * a.Sq + q] = scale * m + __logf(l);
    }
}

// -------------------------------------------------------------------------------- backward (VALU, LDS resident)
// One workgroup per (cloud, head).  Keys/values are processed in LDS-resident chunks of up to 128 rows (one chunk when
// S <= 128); inside a key chunk the queries stream through in chunks of 64 rows (Q, dO and the P/dS tile in LDS).  Every
// thread owns fixed 4x4 micro-tiles of dK and dV of the current key chunk and accumulates them in registers; dQ is
// accumulated across key chunks in global memory by the thread that owns the micro-tile (no atomics: deterministic).
// micro-tile convention: rows {ri + u*RS}, cols {ci + v*CS} with RS/CS = extent/4 (bank-conflict-free strides).
#define ATT_QC 64
#define ATT_KC 128
#define ATT_KT 2            // dK/dV micro-tiles per thread: (KC/4)*(HD/4) <= 256*ATT_KT
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                       const float* __restrict__ dout, const float* __restrict__ lse,
                                                       float* __restrict__ dqkv, int B, int S, int H, float scale, int KR) {
    constexpr int LD = HD + 4;
    constexpr int QC = ATT_QC, RQ = QC / 4, DQ = HD / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDP = KR + 4;                             // KR = key rows per chunk (multiple of 4, <= ATT_KC)
    const int rk = KR / 4;                              // key-side micro-tile stride
    float* Ks = smem;                                   // [KR][LD]
    float* Vs = Ks + (size_t)KR * LD;
    float* Qs = Vs + (size_t)KR * LD;                   // [QC][LD]
    float* Os = Qs + (size_t)QC * LD;                   // [QC][LD]  dO chunk
    float* Ps = Os + (size_t)QC * LD;                   // [QC][LDP] P then dS
    float* Dl = Ps + (size_t)QC * LDP;                  // [QC] D[q]
    float* Ll = Dl + QC;                                // [QC] lse
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int rs = 3 * H * HD, ro = H * HD;
    float* dq_base = dqkv + (size_t)b * S * rs + h * HD;
    const int nkt = rk * DQ;                            // dK / dV micro-tiles of a key chunk

    for (int k0 = 0; k0 < S; k0 += KR) {
        const int kn = min(KR, S - k0);                 // valid key rows in this chunk
        __syncthreads();
        for (int idx = tid; idx < KR * DQ; idx += 256) {
            const int c4 = idx % DQ, row = idx / DQ;
            float4 kx = make_float4(0.f, 0.f, 0.f, 0.f), vx = kx;
            if (row < kn) {
                const float* base = qkv + ((size_t)b * S + k0 + row) * rs + h * HD + c4 * 4;
                kx = *reinterpret_cast<const float4*>(base + H * HD);
                vx = *reinterpret_cast<const float4*>(base + 2 * H * HD);
            }
            *reinterpret_cast<float4*>(Ks + (size_t)row * LD + c4 * 4) = kx;
            *reinterpret_cast<float4*>(Vs + (size_t)row * LD + c4 * 4) = vx;
        }
        float dk[ATT_KT][4][4], dv[ATT_KT][4][4];
#pragma unroll
        for (int t = 0; t < ATT_KT; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) { dk[t][u][v] = 0.f; dv[t][u][v] = 0.f; }

        for (int q0 = 0; q0 < S; q0 += QC) {
            const int qn = min(QC, S - q0);             // valid rows in this query chunk
            __syncthreads();                            // previous chunk fully consumed (and K/V staged)
            for (int idx = tid; idx < QC * DQ; idx += 256) {
                const int c4 = idx % DQ, row = idx / DQ;
                float4 qx = make_float4(0.f, 0.f, 0.f, 0.f), ox = qx;
                if (row < qn) {
                    qx = *reinterpret_cast<const float4*>(qkv + ((size_t)b * S + q0 + row) * rs + h * HD + c4 * 4);
                    ox = *reinterpret_cast<const float4*>(dout + ((size_t)b * S + q0 + row) * ro + h * HD + c4 * 4);
                }
                *reinterpret_cast<float4*>(Qs + (size_t)row * LD + c4 * 4) = qx;
                *reinterpret_cast<float4*>(Os + (size_t)row * LD + c4 * 4) = ox;
            }
            for (int row = wave; row < QC; row += 4) {  // D[q] = sum_d dO[q][d] * O[q][d]
                float acc = 0.f;
                if (row < qn)
                    for (int d = lane; d < HD; d += 64)
                        acc += dout[((size_t)b * S + q0 + row) * ro + h * HD + d] * out[((size_t)b * S + q0 + row) * ro + h * HD + d];
                acc = wave_sum_f32(acc);
                if (lane == 0) { Dl[row] = acc; Ll[row] = row < qn ? lse[((size_t)b * H + h) * S + q0 + row] : 0.f; }
            }
            __syncthreads();
            // ---- P[q][k] = exp(scale * q.k - lse[q])
            for (int mt = tid; mt < RQ * rk; mt += 256) {
                const int ki = mt % rk, qi = mt / rk;
                float acc[4][4] = {};
                for (int d = 0; d < HD; d += 4) {
                    float4 a[4], bb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(Qs + (size_t)(qi + u * RQ) * LD + d);
#pragma unroll
                    for (int v = 0; v < 4; ++v) bb[v] = *reinterpret_cast<const float4*>(Ks + (size_t)(ki + v * rk) * LD + d);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            acc[u][v] += a[u].x * bb[v].x + a[u].y * bb[v].y + a[u].z * bb[v].z + a[u].w * bb[v].w;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int qq = qi + u * RQ, kk = ki + v * rk;
                        Ps[(size_t)qq * LDP + kk] = (qq < qn && kk < kn) ? __expf(scale * acc[u][v] - Ll[qq]) : 0.f;
                    }
            }
            __syncthreads();
            // ---- dV[k][d] += sum_q P[q][k] dO[q][d]
#pragma unroll
            for (int t = 0; t < ATT_KT; ++t) {
                const int mt = tid + 256 * t;
                if (mt < nkt) {
                    const int di = mt % DQ, ki = mt / DQ;
                    for (int qq = 0; qq < qn; ++qq) {
                        float pp[4], g[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) pp[u] = Ps[(size_t)qq * LDP + ki + u * rk];
#pragma unroll
                        for (int v = 0; v < 4; ++v) g[v] = Os[(size_t)qq * LD + di + v * DQ];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int v = 0; v < 4; ++v) dv[t][u][v] += pp[u] * g[v];
                    }
                }
            }
            __syncthreads();
            // ---- dS = P * (dP - D) * scale, dP[q][k] = sum_d dO[q][d] V[k][d]   (in place over P)
            for (int mt = tid; mt < RQ * rk; mt += 256) {
                const int ki = mt % rk, qi = mt / rk;
                float acc[4][4] = {};
                for (int d = 0; d < HD; d += 4) {
                    float4 a[4], bb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(Os + (size_t)(qi + u * RQ) * LD + d);
#pragma unroll
                    for (int v = 0; v < 4; ++v) bb[v] = *reinterpret_cast<const float4*>(Vs + (size_t)(ki + v * rk) * LD + d);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            acc[u][v] += a[u].x * bb[v].x + a[u].y * bb[v].y + a[u].z * bb[v].z + a[u].w * bb[v].w;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int qq = qi + u * RQ, kk = ki + v * rk;
                        const size_t o = (size_t)qq * LDP + kk;
                        Ps[o] = Ps[o] * (acc[u][v] - Dl[qq]) * scale;
                    }
            }
            __syncthreads();
            // ---- dQ[q][d] (+)= sum_k dS[q][k] K[k][d]
            for (int mt = tid; mt < RQ * DQ; mt += 256) {
                const int di = mt % DQ, qi = mt / DQ;
                float acc[4][4] = {};
                for (int kk = 0; kk < kn; ++kk) {
                    float pp[4], g[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) pp[u] = Ps[(size_t)(qi + u * RQ) * LDP + kk];
#pragma unroll
                    for (int v = 0; v < 4; ++v) g[v] = Ks[(size_t)kk * LD + di + v * DQ];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[u][v] += pp[u] * g[v];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int qq = qi + u * RQ;
                    if (qq < qn)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            float* dst = dq_base + (size_t)(q0 + qq) * rs + di + v * DQ;
                            *dst = k0 == 0 ? acc[u][v] : *dst + acc[u][v];
                        }
                }
            }
            // ---- dK[k][d] += sum_q dS[q][k] Q[q][d]
#pragma unroll
            for (int t = 0; t < ATT_KT; ++t) {
                const int mt = tid + 256 * t;
                if (mt < nkt) {
                    const int di = mt % DQ, ki = mt / DQ;
                    for (int qq = 0; qq < qn; ++qq) {
                        float pp[4], g[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) pp[u] = Ps[(size_t)qq * LDP + ki + u * rk];
#pragma unroll
                        for (int v = 0; v < 4; ++v) g[v] = Qs[(size_t)qq * LD + di + v * DQ];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int v = 0; v < 4; ++v) dk[t][u][v] += pp[u] * g[v];
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < ATT_KT; ++t) {
            const int mt = tid + 256 * t;
            if (mt < nkt) {
                const int di = mt % DQ, ki = mt / DQ;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = ki + u * rk;
                    if (kk < kn)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            dq_base[(size_t)(k0 + kk) * rs + H * HD + di + v * DQ] = dk[t][u][v];
                            dq_base[(size_t)(k0 + kk) * rs + 2 * H * HD + di + v * DQ] = dv[t][u][v];
                        }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Attention backward on the matrix cores (fp32-input v_mfma_f32_32x32x2).  One workgroup (4 waves) per (cloud, head); keys come
// from two sources (S0 "prefix" rows of kv0, then the Sq rows of qkv1), so the same kernel serves plain self-attention (S0 = 0)
// and the prompt-prefix attention of the prompt-tuned Transformer (models/dvae.py:536-576).  Per (64-key, 64-query) tile pair:
//   S = Q K^T -> P = exp(scale S - lse) ; dP = dO V^T ; dS = P (dP - D) scale            (wave = one 32x32 quadrant, in regs)
//   dV += P^T dO ; dK += dS^T Q   (accumulated in registers over the query tiles, one [32 keys x 32 dims] quadrant per wave)
//   dQ (+)= dS K                  (accumulated over key tiles through global memory: the workgroup owns its dQ rows)
// Operands of the two head-dimension reductions (S, dP) never touch LDS: the MFMA k-index is free to be any permutation as
// long as A and B agree, so lane half 0 reduces d in [0,HD/2) and half 1 d in [HD/2,HD) and each lane keeps HD/2 CONTIGUOUS
// floats of its Q / dO / K / V row in registers (float4 global loads, K/V fragments live across the query loop).
// The three row reductions take their B operand from row-major LDS tiles Q, dO, K [64][HD] (two consecutive rows land in
// opposite bank halves through an XOR-32 column swizzle) and their A operand from one P/dS tile [64][65].
// Invalid rows are zero-filled and masked out of P; quadrants and reduction ranges made of padding only are skipped.
struct AttnBwdArgs {
    const float* kv0; const float* qkv1; const float* out; const float* dout; const float* lse;
    float* dkv0; float* dqkv1;
    int B, S0, Sq, H; float scale;
};
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int HD, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_bwd_mfma_kernel(const AttnBwdArgs a) {
    constexpr int LDP = 65, DQ = HD / 4, T = 64, FR = HD / 8;      // FR float4 per lane fragment (HD/2 floats)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem; float* Os = Qs + T * HD; float* Ks = Os + T * HD;
    float* Ps = Ks + T * HD; float* Dl = Ps + T * LDP; float* Ll = Dl + T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, khalf = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;               // quadrant of a 64x64 (or 64xHD) tile owned by this wave
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int D = a.H * HD, rs1 = 3 * D, rs0 = 2 * D;
    const int Skv = a.S0 + a.Sq;
    const float* q_base = a.qkv1 + (size_t)b * a.Sq * rs1 + h * HD;
    float* dq_base = a.dqkv1 + (size_t)b * a.Sq * rs1 + h * HD;
    const bool colq = wc * 32 < HD;                         // HD = 32: only the wc = 0 waves own an output quadrant
    // column of this lane inside a row-major [row][HD] tile for the B-operand fetches of rows (2j + khalf)
    const int bcol = HD == 64 ? ((wc * 32 + l32) ^ (32 * khalf)) : l32;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int k0 = 0; k0 < Skv; k0 += T) {
        const int kn = min(T, Skv - k0);
        __syncthreads();                                    // previous key tile fully consumed
        for (int idx = tid; idx < T * DQ; idx += 256) {     // K tile -> LDS (B operand of dQ)
            const int c4 = idx % DQ, row = idx / DQ, kr = k0 + row;
            float4 kx = z4;
            if (row < kn)
                kx = kr < a.S0 ? *reinterpret_cast<const float4*>(a.kv0 + ((size_t)b * a.S0 + kr) * rs0 + h * HD + c4 * 4)
                               : *reinterpret_cast<const float4*>(q_base + (size_t)(kr - a.S0) * rs1 + D + c4 * 4);
            const int col = HD == 64 ? ((c4 * 4) ^ (32 * (row & 1))) : c4 * 4;
            *reinterpret_cast<float4*>(Ks + row * HD + col) = kx;
        }
        // K / V register fragments of this wave's key rows (wc) for the S and dP products
        float4 kf[FR], vf[FR];
        {
            const int row = wc * 32 + l32, kr = k0 + row;
#pragma unroll
            for (int i = 0; i < FR; ++i) { kf[i] = z4; vf[i] = z4; }
            if (row < kn) {
                const float* kb = kr < a.S0 ? a.kv0 + ((size_t)b * a.S0 + kr) * rs0 + h * HD : q_base + (size_t)(kr - a.S0) * rs1 + D;
                const float* vb = kb + D;
#pragma unroll
                for (int i = 0; i < FR; ++i) {
                    kf[i] = *reinterpret_cast<const float4*>(kb + khalf * (HD / 2) + 4 * i);
                    vf[i] = *reinterpret_cast<const float4*>(vb + khalf * (HD / 2) + 4 * i);
                }
            }
        }
        f32x16 dk_acc, dv_acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk_acc[r] = 0.f; dv_acc[r] = 0.f; }

        for (int q0 = 0; q0 < a.Sq; q0 += T) {
            const int qn = min(T, a.Sq - q0);
            __syncthreads();                                // previous query tile fully consumed (and K staged)
            for (int idx = tid; idx < T * DQ; idx += 256) {             // DQ consecutive lanes own one row
                const int c4 = idx % DQ, row = idx / DQ;
                float4 qx = z4, ox = z4, tx = z4;
                if (row < qn) {
                    const size_t o = ((size_t)b * a.Sq + q0 + row) * D + h * HD + c4 * 4;
                    qx = *reinterpret_cast<const float4*>(q_base + (size_t)(q0 + row) * rs1 + c4 * 4);
                    ox = *reinterpret_cast<const float4*>(a.dout + o);
                    tx = *reinterpret_cast<const float4*>(a.out + o);
                }
                const int col = HD == 64 ? ((c4 * 4) ^ (32 * (row & 1))) : c4 * 4;
                *reinterpret_cast<float4*>(Qs + row * HD + col) = qx;
                *reinterpret_cast<float4*>(Os + row * HD + col) = ox;
                // D[q] = sum_d dO[q][d] * O[q][d]: reduce the DQ per-lane partials inside the 16-lane DPP row
                float part = ox.x * tx.x + ox.y * tx.y + ox.z * tx.z + ox.w * tx.w;
                part += dpp_mov_f<0x111, 0xf>(0.f, part);
                part += dpp_mov_f<0x112, 0xf>(0.f, part);
                part += dpp_mov_f<0x114, 0xf>(0.f, part);
                if (DQ == 16) part += dpp_mov_f<0x118, 0xf>(0.f, part);
                if (c4 == DQ - 1) Dl[row] = part;
                if (c4 == 0) Ll[row] = row < qn ? a.lse[((size_t)b * a.H + h) * a.Sq + q0 + row] : 0.f;
            }
            // ---- S and dP quadrants [32 queries (wr) x 32 keys (wc)], reduction over the head dimension from registers
            f32x16 s_acc, p_acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s_acc[r] = 0.f; p_acc[r] = 0.f; }
            if (wr * 32 < qn && wc * 32 < kn) {             // quadrants made of padding only stay zero
                const int row = wr * 32 + l32;
                float4 qf[FR], of[FR];
#pragma unroll
                for (int i = 0; i < FR; ++i) { qf[i] = z4; of[i] = z4; }
                if (row < qn) {
                    const float* qb = q_base + (size_t)(q0 + row) * rs1 + khalf * (HD / 2);
                    const float* ob = a.dout + ((size_t)b * a.Sq + q0 + row) * D + h * HD + khalf * (HD / 2);
#pragma unroll
                    for (int i = 0; i < FR; ++i) {
                        qf[i] = *reinterpret_cast<const float4*>(qb + 4 * i);
                        of[i] = *reinterpret_cast<const float4*>(ob + 4 * i);
                    }
                }
#pragma unroll
                for (int i = 0; i < FR; ++i) {
                    s_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[i].x, kf[i].x, s_acc, 0, 0, 0);
                    p_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(of[i].x, vf[i].x, p_acc, 0, 0, 0);
                    s_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[i].y, kf[i].y, s_acc, 0, 0, 0);
                    p_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(of[i].y, vf[i].y, p_acc, 0, 0, 0);
                    s_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[i].z, kf[i].z, s_acc, 0, 0, 0);
                    p_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(of[i].z, vf[i].z, p_acc, 0, 0, 0);
                    s_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[i].w, kf[i].w, s_acc, 0, 0, 0);
                    p_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(of[i].w, vf[i].w, p_acc, 0, 0, 0);
                }
            }
            __syncthreads();                                // Q / dO tiles, D and lse visible
            const int kk = wc * 32 + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const float p = (qq < qn && kk < kn) ? __expf(a.scale * s_acc[r] - Ll[qq]) : 0.f;
                Ps[qq * LDP + kk] = p;
                s_acc[r] = p * (p_acc[r] - Dl[qq]) * a.scale;          // dS, kept in registers until P has been consumed
            }
            __syncthreads();
            const int qn2 = (qn + 7) & ~7, kn2 = (kn + 7) & ~7;       // reductions stop at the valid rows (zero padding beyond)
            // ---- dV[k][d] += sum_q P[q][k] dO[q][d]     quadrant: keys wr, dims wc
            if (colq && wr * 32 < kn) {
                const float* pa = Ps + khalf * LDP + wr * 32 + l32;
                const float* ob = Os + khalf * HD + bcol;
#pragma unroll 4
                for (int q = 0; q < qn2; q += 2)
                    dv_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[q * LDP], ob[q * HD], dv_acc, 0, 0, 0);
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                Ps[qq * LDP + kk] = s_acc[r];
            }
            __syncthreads();
            if (colq && wr * 32 < kn) {
                // ---- dK[k][d] += sum_q dS[q][k] Q[q][d]
                const float* pa = Ps + khalf * LDP + wr * 32 + l32;
                const float* qb = Qs + khalf * HD + bcol;
#pragma unroll 4
                for (int q = 0; q < qn2; q += 2)
                    dk_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[q * LDP], qb[q * HD], dk_acc, 0, 0, 0);
            }
            if (colq && wr * 32 < qn) {
                // ---- dQ[q][d] (+)= sum_k dS[q][k] K[k][d]   quadrant: queries wr, dims wc
                f32x16 dq_acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) dq_acc[r] = 0.f;
                const float* sa = Ps + (wr * 32 + l32) * LDP + khalf;
                const float* kb2 = Ks + khalf * HD + bcol;
#pragma unroll 4
                for (int k = 0; k < kn2; k += 2)
                    dq_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[k], kb2[k * HD], dq_acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qq = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    if (qq < qn) {
                        float* dst = dq_base + (size_t)(q0 + qq) * rs1 + wc * 32 + l32;
                        *dst = k0 == 0 ? dq_acc[r] : *dst + dq_acc[r];
                    }
                }
            }
        }
        if (colq) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kq = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf, kr = k0 + kq;
                if (kq < kn) {
                    if (kr < a.S0) {
                        float* dst = a.dkv0 + ((size_t)b * a.S0 + kr) * rs0 + h * HD + wc * 32 + l32;
                        dst[0] = dk_acc[r]; dst[D] = dv_acc[r];
                    } else {
                        float* dst = dq_base + (size_t)(kr - a.S0) * rs1 + wc * 32 + l32;
                        dst[D] = dk_acc[r]; dst[2 * D] = dv_acc[r];
                    }
                }
            }
        }
    }
}


// =================================================================================== S <= 16: one wave per (cloud, head), registers only
// The student encoder of Stage II runs on 13 visible tokens + cls = 14 (models/act.py:255): a 32- or 64-wide tile is 80-95 % padding and
// the LDS-staged kernels above are pure latency (7 % MFMA utilisation in the backward).  Here one wave owns one (cloud, head) pair; every
// product is a handful of v_mfma_f32_16x16x4_f32 whose operands come straight from global memory into registers -- no LDS, no barrier:
//   operand "form R" : lane (r = lane&15, g = lane>>4) holds X[r][E*g .. E*g+E-1]  (E = HD/4 contiguous floats, MFMA k-step i uses element i:
//                      the reduction over the head dimension is order-free, so both operands simply use the same permutation)
//   operand "form C" : lane (c, g) holds X[4g + s][16*blk + c] for s = 0..3 (k-step s reduces over rows 4g+s: again a free permutation)
//   MFMA C/D layout  : lane (c, g) holds D[4g + reg][c]
// forward : St = K Q^t (D: key = 4g+reg, query = c) -> softmax over keys = in-lane + two lane^16/32 exchanges -> Pt is directly the
//           B operand of Ot = Vt Pt (key = 4g+s at k-step s) with V in form C; each lane stores 4 consecutive output floats.
// backward: S = Q K^t and St = K Q^t (P and Pt), dP = dO V^t and dPt = V dO^t, dS = P (dP - delta) (both orientations) and then
//           dVt = dOt P, dKt = Qt dS, dQt = Kt dSt with P / dS / dSt as B operands straight from their D registers.
template <int HD>
__device__ __forceinline__ void load_form_r(const float* __restrict__ base, int ld, int r, int g, int S, float* x) {
    constexpr int E = HD / 4;
    if (r < S) {
        const float4* p = reinterpret_cast<const float4*>(base + (size_t)r * ld + E * g);
#pragma unroll
        for (int i = 0; i < E / 4; ++i) { const float4 t = p[i]; x[4 * i] = t.x; x[4 * i + 1] = t.y; x[4 * i + 2] = t.z; x[4 * i + 3] = t.w; }
    } else {
#pragma unroll
        for (int i = 0; i < E; ++i) x[i] = 0.f;
    }
}
template <int HD>
__device__ __forceinline__ void load_form_c(const float* __restrict__ base, int ld, int c, int g, int S, float (*x)[HD / 16]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int row = 4 * g + s;
#pragma unroll
        for (int blk = 0; blk < HD / 16; ++blk) x[s][blk] = row < S ? base[(size_t)row * ld + 16 * blk + c] : 0.f;
    }
}
typedef float f32x4s __attribute__((ext_vector_type(4)));

template <int HD>
__global__ __launch_bounds__(256) void attn_small_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse,
                                                             int B, int S, int H, float scale) {
    constexpr int E = HD / 4, NB = HD / 16;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long long pair = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= (long long)B * H) return;
    const int b = (int)(pair / H), h = (int)(pair % H);
    const int D = H * HD, ld = 3 * D;
    const float* q0 = qkv + (size_t)b * S * ld + h * HD;
    float qr[E], kr[E], vc[4][NB];
    load_form_r<HD>(q0, ld, c, g, S, qr);
    load_form_r<HD>(q0 + D, ld, c, g, S, kr);
    load_form_c<HD>(q0 + 2 * D, ld, c, g, S, vc);
    f32x4s st = {0.f, 0.f, 0.f, 0.f};                                  // St[key = 4g+reg][query = c]
#pragma unroll
    for (int i = 0; i < E; ++i) st = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[i], qr[i], st, 0, 0, 0);
    float m = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { if (4 * g + r >= S) st[r] = -3.0e38f; m = fmaxf(m, st[r]); }
    m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { st[r] = __expf(scale * (st[r] - m)); l += st[r]; }
    l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
    const float inv_l = 1.0f / l;
    float* op = out + ((size_t)b * S + min(c, S - 1)) * D + h * HD;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {                               // Ot[d = 16 blk + 4g + reg][query = c]; MFMAs outside divergent control flow
        f32x4s o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) o = __builtin_amdgcn_mfma_f32_16x16x4f32(vc[s2][blk], st[s2], o, 0, 0, 0);
        if (c < S) *reinterpret_cast<float4*>(op + 16 * blk + 4 * g) = make_float4(o[0] * inv_l, o[1] * inv_l, o[2] * inv_l, o[3] * inv_l);
    }
    if (lse && g == 0 && c < S) lse[((size_t)b * H + h) * S + c] = scale * m + __logf(l);
}

template <int HD>
__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ out, const float* __restrict__ dout,
                                                             const float* __restrict__ lse, float* __restrict__ dqkv, int B, int S, int H, float scale) {
    constexpr int E = HD / 4, NB = HD / 16;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long long pair = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= (long long)B * H) return;
    const int b = (int)(pair / H), h = (int)(pair % H);
    const int D = H * HD, ld = 3 * D;
    const float* q0 = qkv + (size_t)b * S * ld + h * HD;
    const float* o0 = out + (size_t)b * S * D + h * HD;
    const float* do0 = dout + (size_t)b * S * D + h * HD;
    float* dq0 = dqkv + (size_t)b * S * ld + h * HD;
    float qr[E], kr[E], vr[E], dor[E], orr[E];
    load_form_r<HD>(q0, ld, c, g, S, qr);
    load_form_r<HD>(q0 + D, ld, c, g, S, kr);
    load_form_r<HD>(q0 + 2 * D, ld, c, g, S, vr);
    load_form_r<HD>(do0, D, c, g, S, dor);
    load_form_r<HD>(o0, D, c, g, S, orr);
    // delta[row c] = sum_d dO[c][d] O[c][d]; log-sum-exp of row c
    float delta = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) delta += dor[i] * orr[i];
    delta += __shfl_xor(delta, 16); delta += __shfl_xor(delta, 32);
    const float lse_c = c < S ? lse[((size_t)b * H + h) * S + c] : 0.f;
    // S / St and dP / dPt
    f32x4s s_qk = {0.f, 0.f, 0.f, 0.f}, s_kq = s_qk, dp = s_qk, dpt = s_qk;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        s_qk = __builtin_amdgcn_mfma_f32_16x16x4f32(qr[i], kr[i], s_qk, 0, 0, 0);      // S [query = 4g+reg][key = c]
        s_kq = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[i], qr[i], s_kq, 0, 0, 0);      // St[key = 4g+reg][query = c]
        dp   = __builtin_amdgcn_mfma_f32_16x16x4f32(dor[i], vr[i], dp, 0, 0, 0);       // dP [query = 4g+reg][key = c]
        dpt  = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[i], dor[i], dpt, 0, 0, 0);      // dPt[key = 4g+reg][query = c]
    }
    f32x4s p, ds, dst;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        // orientation 1: query = row, key = c   (needs lse / delta of the ROW query: fetched from the lane that owns it)
        const float lse_r = __shfl(lse_c, row), del_r = __shfl(delta, row);
        const bool v1 = row < S && c < S;
        p[r] = v1 ? __expf(scale * s_qk[r] - lse_r) : 0.f;
        ds[r] = p[r] * (dp[r] - del_r) * scale;
        // orientation 2: key = row, query = c
        const bool v2 = row < S && c < S;
        const float pt = v2 ? __expf(scale * s_kq[r] - lse_c) : 0.f;
        dst[r] = pt * (dpt[r] - delta) * scale;
    }
    float xc[4][NB];
    // dVt[d][key = c] = sum_q dOt[d][q] P[q][key]      (A: dO form C, B: P)
    load_form_c<HD>(do0, D, c, g, S, xc);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        f32x4s a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) a = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[s2][blk], p[s2], a, 0, 0, 0);
        if (c < S) *reinterpret_cast<float4*>(dq0 + 2 * D + (size_t)c * ld + 16 * blk + 4 * g) = make_float4(a[0], a[1], a[2], a[3]);
    }
    // dKt[d][key = c] = sum_q Qt[d][q] dS[q][key]
    load_form_c<HD>(q0, ld, c, g, S, xc);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        f32x4s a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) a = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[s2][blk], ds[s2], a, 0, 0, 0);
        if (c < S) *reinterpret_cast<float4*>(dq0 + D + (size_t)c * ld + 16 * blk + 4 * g) = make_float4(a[0], a[1], a[2], a[3]);
    }
    // dQt[d][query = c] = sum_key Kt[d][key] dSt[key][query]
    load_form_c<HD>(q0 + D, ld, c, g, S, xc);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        f32x4s a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) a = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[s2][blk], dst[s2], a, 0, 0, 0);
        if (c < S) *reinterpret_cast<float4*>(dq0 + (size_t)c * ld + 16 * blk + 4 * g) = make_float4(a[0], a[1], a[2], a[3]);
    }
}


// =================================================================================== register-resident kernels for ANY sequence length (round 3)
// One wave owns a 32-row block of one (cloud, head) and streams the other side through in 32-row tiles.  No LDS, no barrier: every operand
// goes from global memory (L1 / L2 hits: a (cloud, head) pair's K, V, Q, dO are 16-32 KB each) straight into the register form its MFMA wants --
//   "row form"  lane (row = lane&31, half = lane>>5) holds X[row][half*HD/2 .. +HD/2)   (A or B operand of a head-dimension reduction: the
//               reduction index of an MFMA is a free permutation as long as A and B agree, so the two halves split the head dimension)
//   "col form"  lane (c = lane&31, half) holds X[f(r, half)][dt*32 + c], f(r, half) = (r&3) + 8*(r>>2) + 4*half   (A operand of a reduction over
//               rows: row f(r, half) is exactly the row the C/D register r of that half-wave belongs to, so P / dS / dS^t are B operands
//               straight from their accumulator registers)
// Forward: S^t = K Q^t (lane = one query: softmax in registers + one lane^32 exchange), O^t += V^t P^t with an online softmax per 32-key tile.
// Backward, two roles in ONE launch (no workspace, no ordering between them; both recompute P from the saved log-sum-exp):
//   dQ role   (one wave per 32 queries):  S^t, dP^t = V dO^t, dS^t = P^t (dP^t - D) scale, dQ^t += K^t dS^t            96 MFMAs per 32x32 tile pair
//   dK/dV role (one wave per 32 keys):    S = Q K^t, dP = dO V^t, dV^t += dO^t P, dK^t += Q^t dS                         128 MFMAs per tile pair
// (160 would do with a shared S / dP; the price of the 224 is what buys a barrier-free, LDS-free, occupancy-2 kernel whose MFMA pipe stays
// busy -- the workgroup-per-pair kernel above sits at 14 % MFMA utilisation behind its five barriers per tile pair.)
// Keys come from two row segments (S0 prefix rows of kv0, then the S1 rows of the packed qkv1), as in the kernels above.
struct AttnRegArgs {
    const float* q;  const float* k0; const float* v0; const float* k1; const float* v1;      // head 0 of row 0 of each operand
    const float* out; const float* dout; const float* lse;                                    // backward only
    float* dq; float* dk0; float* dv0; float* dk1; float* dv1;                                // backward outputs (same strides as the inputs)
    float* o; float* lse_out;                                                                 // forward outputs
    long long q_bs, kv0_bs, kv1_bs;      // per-cloud strides (floats)
    int ldq, ld0, ld1;                   // row strides (floats)
    int B, H, Sq, S0, S1;
    float scale;
};
#define ATT_F(r, half) (((r) & 3) + 8 * ((r) >> 2) + 4 * (half))

// Tails without clamps: the LAST 32-row tile of a side is shifted back to end exactly at the last row (rows S-32 .. S-1, all valid) and the
// rows it shares with the previous tile are masked out of P (keys) or simply not stored (owned rows).  So every tile is a full tile inside ONE
// segment (host-side condition: S0 % 32 == 0, S1 >= 32, Sq >= 32), every row address is a wave-uniform base + a 32-bit lane offset, and a
// col-form fetch is a plain run of global_load_dword with SGPR bases.
struct AttSeg { const float* base; int ld; };                           // rows t0 .. t0+31 of a two-segment operand (wave-uniform)
__device__ __forceinline__ AttSeg att_seg_tile(const float* __restrict__ p0, int ld0, const float* __restrict__ p1, int ld1, int S0, int t0) {
    AttSeg g;
    if (t0 >= S0) { g.base = p1 + (size_t)(t0 - S0) * ld1; g.ld = ld1; } else { g.base = p0 + (size_t)t0 * ld0; g.ld = ld0; }
    return g;
}
template <int HD>
__device__ __forceinline__ void att_load_row_form(const AttSeg g, int row, int half, float* x) {              // lane's own row of the tile
    const unsigned off = (unsigned)(row * g.ld + half * (HD / 2));
#pragma unroll
    for (int i = 0; i < HD / 8; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(g.base + off + 4 * i);
        x[4 * i] = t.x; x[4 * i + 1] = t.y; x[4 * i + 2] = t.z; x[4 * i + 3] = t.w;
    }
}
template <int HD>
__device__ __forceinline__ void att_load_col_form(const AttSeg g, int c, int half, float (*x)[HD / 32]) {
    // The m index of the products that consume a col form (the head dimension d of O^t, dQ^t, dK^t, dV^t) is free to be any permutation as long as
    // the store agrees: lane c holds d = (HD/32) * c + dt, so its HD/32 values per row are CONTIGUOUS -- one dwordx2 load per row and lane pair of
    // accumulators instead of two dword loads 128 bytes apart (global-load instructions per tile are what these kernels are short of).
    constexpr int NDT = HD / 32;
    const unsigned lane_off = (unsigned)(4 * half * g.ld + NDT * c);     // per-lane part; the row part below is wave-uniform
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float* rb = g.base + (size_t)ATT_F(r, 0) * g.ld;
        if constexpr (NDT == 2) {
            const float2 v = *reinterpret_cast<const float2*>(rb + lane_off);
            x[r][0] = v.x; x[r][1] = v.y;
        } else {
            x[r][0] = rb[lane_off];
        }
    }
}
// store an accumulator set in the o-layout (acc[dt][r] = X^t[d = (HD/32) * f(r, half) + dt][row = lane&31]) as row-major X[row][d], scaled
template <int HD>
__device__ __forceinline__ void att_store_o(float* __restrict__ rowp, int half, const f32x16* acc, float mul) {
    constexpr int NDT = HD / 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) {                                        // registers 4g .. 4g+3 = rows m = 8g + 4 half + (0..3) -> NDT * 4 consecutive d
        if constexpr (NDT == 2) {
            float4 t0, t1;
            t0.x = acc[0][g * 4 + 0] * mul; t0.y = acc[1][g * 4 + 0] * mul; t0.z = acc[0][g * 4 + 1] * mul; t0.w = acc[1][g * 4 + 1] * mul;
            t1.x = acc[0][g * 4 + 2] * mul; t1.y = acc[1][g * 4 + 2] * mul; t1.z = acc[0][g * 4 + 3] * mul; t1.w = acc[1][g * 4 + 3] * mul;
            *reinterpret_cast<float4*>(rowp + 16 * g + 8 * half) = t0;
            *reinterpret_cast<float4*>(rowp + 16 * g + 8 * half + 4) = t1;
        } else {
            float4 t;
            t.x = acc[0][g * 4 + 0] * mul; t.y = acc[0][g * 4 + 1] * mul; t.z = acc[0][g * 4 + 2] * mul; t.w = acc[0][g * 4 + 3] * mul;
            *reinterpret_cast<float4*>(rowp + 8 * g + 4 * half) = t;
        }
    }
}
// (wave-uniform work-item decomposition; 32-bit divisions expand to VALU code, so the results are pinned back into SGPRs)
__device__ __forceinline__ void att_item(unsigned item, int ntiles, int H, int& tile, int& b, int& h) {
    tile = __builtin_amdgcn_readfirstlane((int)(item % (unsigned)ntiles));
    const unsigned pr = item / (unsigned)ntiles;
    b = __builtin_amdgcn_readfirstlane((int)(pr / (unsigned)H));
    h = __builtin_amdgcn_readfirstlane((int)(pr % (unsigned)H));
}

// Fetch placement (measured on the Stage-I prefix shape, 128 clouds x 12 heads x (64 q x 128 k), and on S = 512): every fragment is fetched
// right where it is consumed and the co-resident waves of the SIMD cover the latency.  Issuing the fetches one MFMA burst early (software
// pipelining: +32 ... +64 registers, one wave per SIMD fewer) was 15-25 % SLOWER, and parking coalesced 16-byte loads in a wave-private LDS
// slab to pick both fragment forms out of it (8 instead of 40 global loads per tile) was 70-90 % slower: at these sizes the call moves about as
// many HBM bytes as it has matrix work (Stage-I shape: 150 MB = 25 us at 6 TB/s against 20.5 us of MFMA issue; with every in-loop fetch removed
// the forward still takes 44.8 us, i.e. the Q / O prologue and epilogue bursts of a single resident round of waves do not overlap the matrix
// work), so what counts is waves in flight, not the instruction stream of one wave.
template <int HD>
__global__ __launch_bounds__(256, 3) void attn_fwd_reg_kernel(const AttnRegArgs a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ql = lane & 31, half = lane >> 5;
    const int nqt = (a.Sq + 31) >> 5, Sk = a.S0 + a.S1;
    const unsigned item = blockIdx.x * 4u + (unsigned)wave;
    if (item >= (unsigned)(a.B * a.H * nqt)) return;
    int qt, b, h;
    att_item(item, nqt, a.H, qt, b, h);
    const int q0n = qt * 32, q0 = min(q0n, a.Sq - 32);                  // nominal / shifted first query of this wave
    const float* k0 = a.k0 + (size_t)b * a.kv0_bs + h * HD; const float* v0 = a.v0 + (size_t)b * a.kv0_bs + h * HD;
    const float* k1 = a.k1 + (size_t)b * a.kv1_bs + h * HD; const float* v1 = a.v1 + (size_t)b * a.kv1_bs + h * HD;

    float qreg[HD / 2];
    att_load_row_form<HD>(AttSeg{a.q + (size_t)b * a.q_bs + (size_t)q0 * a.ldq + h * HD, a.ldq}, ql, half, qreg);
    f32x16 o[HD / 32];
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = -3.0e38f, l = 0.f;
    const float scale = a.scale;

    for (int t0n = 0; t0n < Sk; t0n += 32) {
        const int t0 = min(t0n, Sk - 32);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            float kf[HD / 2];
            att_load_row_form<HD>(att_seg_tile(k0, a.ld0, k1, a.ld1, a.S0, t0), ql, half, kf);
#pragma unroll
            for (int s = 0; s < HD / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qreg[s], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                              // the V fragment re-uses the K fragment's registers: its loads are in flight
        float vf[16][HD / 32];                                          // during the softmax; the other waves of the SIMD own the MFMA pipe meanwhile
        att_load_col_form<HD>(att_seg_tile(v0, a.ld0, v1, a.ld1, a.S0, t0), ql, half, vf);
        float mc = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = (t0 + ATT_F(r, half) >= t0n) ? acc[r] : -3.0e38f;     // rows shared with the previous tile are masked
            acc[r] = v;
            mc = fmaxf(mc, v);
        }
        mc = fmaxf(mc, __shfl_xor(mc, 32));
        const float mn = fmaxf(m, mc);
        const float alpha = __expf(scale * (m - mn));                   // 0 on the first tile (m = -huge)
        m = mn;
        float lc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __expf(scale * (acc[r] - m));               // masked keys: exp(-huge) == 0
            acc[r] = p;
            lc += p;
        }
        lc += __shfl_xor(lc, 32);
        l = l * alpha + lc;
        if (t0n > 0) {
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r][dt], acc[r], o[dt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int q = q0 + ql;
    if (q >= q0n) {
        att_store_o<HD>(a.o + ((size_t)b * a.Sq + q) * (a.H * HD) + h * HD, half, o, 1.0f / l);
        if (a.lse_out && half == 0) a.lse_out[((size_t)b * a.H + h) * a.Sq + q] = scale * m + __logf(l);
    }
}

template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_reg_kernel(const AttnRegArgs a, int n_dq_blocks) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ql = lane & 31, half = lane >> 5;
    const int Sk = a.S0 + a.S1, D = a.H * HD;
    const int nqt = (a.Sq + 31) >> 5, nkt = (Sk + 31) >> 5;
    const float scale = a.scale;
    if ((int)blockIdx.x < n_dq_blocks) {
        // ------------------------------------------------------------------ dQ role: this wave owns 32 queries
        const unsigned item = blockIdx.x * 4u + (unsigned)wave;
        if (item >= (unsigned)(a.B * a.H * nqt)) return;
        int qt, b, h;
        att_item(item, nqt, a.H, qt, b, h);
        const int q0n = qt * 32, q0 = min(q0n, a.Sq - 32);
        const float* k0 = a.k0 + (size_t)b * a.kv0_bs + h * HD; const float* v0 = a.v0 + (size_t)b * a.kv0_bs + h * HD;
        const float* k1 = a.k1 + (size_t)b * a.kv1_bs + h * HD; const float* v1 = a.v1 + (size_t)b * a.kv1_bs + h * HD;
        float qreg[HD / 2], dor[HD / 2];
        att_load_row_form<HD>(AttSeg{a.q + (size_t)b * a.q_bs + (size_t)q0 * a.ldq + h * HD, a.ldq}, ql, half, qreg);
        att_load_row_form<HD>(AttSeg{a.dout + ((size_t)b * a.Sq + q0) * D + h * HD, D}, ql, half, dor);
        float dlt;
        {
            float orow[HD / 2];
            att_load_row_form<HD>(AttSeg{a.out + ((size_t)b * a.Sq + q0) * D + h * HD, D}, ql, half, orow);
            float part = 0.f;
#pragma unroll
            for (int s = 0; s < HD / 2; ++s) part += dor[s] * orow[s];
            dlt = part + __shfl_xor(part, 32);                          // D[q] = sum_d dO[q][d] O[q][d]
        }
        const float lq = a.lse[((size_t)b * a.H + h) * a.Sq + q0 + ql];
        f32x16 dq[HD / 32];
#pragma unroll
        for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
        for (int t0n = 0; t0n < Sk; t0n += 32) {
            const int t0 = min(t0n, Sk - 32);
            f32x16 st, dpt;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
            const AttSeg kg = att_seg_tile(k0, a.ld0, k1, a.ld1, a.S0, t0);
            {
                float kf[HD / 2], vf[HD / 2];
                att_load_row_form<HD>(kg, ql, half, kf);
                att_load_row_form<HD>(att_seg_tile(v0, a.ld0, v1, a.ld1, a.S0, t0), ql, half, vf);
#pragma unroll
                for (int s = 0; s < HD / 2; ++s) {
                    st  = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qreg[s], st, 0, 0, 0);
                    dpt = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[s], dor[s], dpt, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                          // K in col form re-uses the fragment registers; in flight during the dS arithmetic
            float kc[16][HD / 32];
            att_load_col_form<HD>(kg, ql, half, kc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = (t0 + ATT_F(r, half) >= t0n) ? __expf(scale * st[r] - lq) : 0.f;
                st[r] = p * (dpt[r] - dlt) * scale;                     // dS^t[key = f(r, half)][query = ql]
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int dt = 0; dt < HD / 32; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[r][dt], st[r], dq[dt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int q = q0 + ql;
        if (q >= q0n) att_store_o<HD>(a.dq + (size_t)b * a.q_bs + (size_t)q * a.ldq + h * HD, half, dq, 1.0f);
        return;
    }
    // ---------------------------------------------------------------------- dK / dV role: this wave owns 32 keys
    const unsigned item = (blockIdx.x - (unsigned)n_dq_blocks) * 4u + (unsigned)wave;
    if (item >= (unsigned)(a.B * a.H * nkt)) return;
    int kt, b, h;
    att_item(item, nkt, a.H, kt, b, h);
    const int k0n = kt * 32, k0s = min(k0n, Sk - 32);
    float kreg[HD / 2], vreg[HD / 2];
    att_load_row_form<HD>(att_seg_tile(a.k0 + (size_t)b * a.kv0_bs + h * HD, a.ld0, a.k1 + (size_t)b * a.kv1_bs + h * HD, a.ld1, a.S0, k0s), ql, half, kreg);
    att_load_row_form<HD>(att_seg_tile(a.v0 + (size_t)b * a.kv0_bs + h * HD, a.ld0, a.v1 + (size_t)b * a.kv1_bs + h * HD, a.ld1, a.S0, k0s), ql, half, vreg);
    const float* qb = a.q + (size_t)b * a.q_bs + h * HD;
    const float* ob = a.out + (size_t)b * a.Sq * D + h * HD;
    const float* gb = a.dout + (size_t)b * a.Sq * D + h * HD;
    const float* lb = a.lse + ((size_t)b * a.H + h) * a.Sq;
    f32x16 dk[HD / 32], dv[HD / 32];
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    for (int u0n = 0; u0n < a.Sq; u0n += 32) {
        const int u0 = min(u0n, a.Sq - 32);
        const AttSeg qg{qb + (size_t)u0 * a.ldq, a.ldq}, gg{gb + (size_t)u0 * D, D};
        float dlt, lq = lb[u0 + ql];
        f32x16 sa, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
        {
            float gf[HD / 2];
            att_load_row_form<HD>(gg, ql, half, gf);
            {
                float orow[HD / 2];
                att_load_row_form<HD>(AttSeg{ob + (size_t)u0 * D, D}, ql, half, orow);
                float part = 0.f;
#pragma unroll
                for (int s = 0; s < HD / 2; ++s) part += gf[s] * orow[s];
                dlt = part + __shfl_xor(part, 32);                      // lane (ql, *) holds D and lse of query u0 + ql
            }
#pragma unroll
            for (int s = 0; s < HD / 2; ++s) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(gf[s], vreg[s], dp, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            float qf[HD / 2];
            att_load_row_form<HD>(qg, ql, half, qf);
#pragma unroll
            for (int s = 0; s < HD / 2; ++s) sa = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[s], kreg[s], sa, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        float gc[16][HD / 32];
        att_load_col_form<HD>(gg, ql, half, gc);                        // dO, col form
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int src = ATT_F(r, half);                             // register r belongs to query u0 + f(r, half): fetch its lse and D
            const float lr = __shfl(lq, src), dr = __shfl(dlt, src);
            const float p = (u0 + src >= u0n) ? __expf(scale * sa[r] - lr) : 0.f;      // queries shared with the previous tile are masked
            sa[r] = p;
            dp[r] = p * (dp[r] - dr) * scale;                           // dS[query = f(r, half)][key = ql]
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(gc[r][dt], sa[r], dv[dt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        att_load_col_form<HD>(qg, ql, half, gc);                        // Q, col form
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(gc[r][dt], dp[r], dk[dt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int key = k0s + ql;
    if (key >= k0n) {
        float* dkp = key < a.S0 ? a.dk0 + (size_t)b * a.kv0_bs + (size_t)key * a.ld0 + h * HD : a.dk1 + (size_t)b * a.kv1_bs + (size_t)(key - a.S0) * a.ld1 + h * HD;
        float* dvp = key < a.S0 ? a.dv0 + (size_t)b * a.kv0_bs + (size_t)key * a.ld0 + h * HD : a.dv1 + (size_t)b * a.kv1_bs + (size_t)(key - a.S0) * a.ld1 + h * HD;
        att_store_o<HD>(dkp, half, dk, 1.0f);
        att_store_o<HD>(dvp, half, dv, 1.0f);
    }
}

// =================================================================================== single-pass backward (round 6)
// The two-role kernel above executes 7 matmul units per 5 algorithmic ones (S and dP are recomputed in both roles) and fetches every fragment of
// every tile pair from global memory.  This kernel computes S and dP ONCE per 32x32 tile pair: 160 MFMAs instead of 224.
//   * a workgroup = 4 waves owns PAIRS = 4 / KW (cloud, head) pairs; wave kw of a pair owns 32 keys of the current block of KW key tiles: K, V rows stay in
//     registers (row form) and dK^t, dV^t accumulate in registers over all query tiles, exactly like the dK / dV role above;
//   * the 32-query tile of Q and dO is staged ONCE per pair in LDS ([32][HD + 4]; both fragment forms are conflict-free reads of it: row form = b128,
//     col form = b64) together with lse and D = rowsum(dO * O) of its queries, instead of 40 global loads per wave and tile pair;
//   * dS goes through a wave-private [32][36] LDS tile to come back transposed (4 x ds_write_b128 + 16 x ds_read_b32, no barrier: the wave's own LDS
//     queue is in order), which makes dS^t the B operand of the wave's partial dQ^t = K^t dS^t over ITS 32 keys (K in col form: the only global fragment);
//   * the KW partial dQ tiles of a pair meet in LDS and are summed in wave order by the pair's threads (deterministic, no atomics); with more than one
//     key block the sum continues through global memory (the workgroup owns its rows).
// Two workgroup barriers per query tile (stage -> compute -> reduce); two workgroups per CU cover each other's staging.  Tails as above: the last
// tile of a side is shifted back to end at the last row, the rows it shares with its neighbour are masked out of P / not stored.
// NP = pairs per workgroup (64 * KW * NP threads): 4 / KW fills a 256-thread workgroup; NP = 1 at KW = 2 (S = 64: 128-thread workgroups, one pair each) spreads the
// 768 pairs of the student decoder over three workgroups per CU instead of 1.5 (A/B: ACT_ATTN_BWD_NP)
template <int HD, int KW, int NP = 4 / KW>
__global__ __launch_bounds__(64 * KW * NP, 2) void attn_bwd_one_kernel(const AttnRegArgs a) {
    constexpr int PAIRS = NP, NDT = HD / 32, LDQ = HD + 4, LDT = 36, C4 = HD / 4;
    constexpr int PSZ = 2 * 32 * LDQ + 64;                             // floats per pair: Q tile | dO tile | lse[32] | D[32]
    constexpr int XSZ = 32 * LDQ;                                      // floats per wave: dS^t [32][LDT] first, then the partial dQ [32][LDQ]
    constexpr int NT = 64 * KW;                                        // threads of a pair group
    constexpr int ITS = 32 * C4 / NT;                                  // float4 per thread and staged operand
    static_assert(32 * LDT <= XSZ && (32 * C4) % NT == 0, "layout");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, half = lane >> 5;
    const int pl = wave / KW, kw = wave % KW, tp = tid - pl * NT;      // pair slot, key tile inside the block, thread inside the pair group
    const int Sk = a.S0 + a.S1, D = a.H * HD;
    const int nqt = (a.Sq + 31) >> 5, nkt = (Sk + 31) >> 5, nkb = (nkt + KW - 1) / KW;
    const float scale = a.scale;
    const unsigned pr = blockIdx.x * (unsigned)PAIRS + (unsigned)pl;
    const bool pair_ok = pr < (unsigned)(a.B * a.H);
    const int b = __builtin_amdgcn_readfirstlane(pair_ok ? (int)(pr / (unsigned)a.H) : 0);
    const int h = __builtin_amdgcn_readfirstlane(pair_ok ? (int)(pr % (unsigned)a.H) : 0);
    float* Qs = smem + (size_t)pl * PSZ; float* Gs = Qs + 32 * LDQ; float* Ls = Gs + 32 * LDQ; float* Ds = Ls + 32;
    float* Xp = smem + (size_t)PAIRS * PSZ + (size_t)pl * KW * XSZ;    // the KW partial tiles of this pair
    float* X = Xp + (size_t)kw * XSZ;                                  // this wave's
    const float* qb = a.q + (size_t)b * a.q_bs + h * HD;
    const float* ob = a.out + (size_t)b * a.Sq * D + h * HD;
    const float* gb = a.dout + (size_t)b * a.Sq * D + h * HD;
    const float* lb = a.lse + ((size_t)b * a.H + h) * a.Sq;
    float* dqb = a.dq + (size_t)b * a.q_bs + h * HD;
    const float* k0 = a.k0 + (size_t)b * a.kv0_bs + h * HD; const float* v0 = a.v0 + (size_t)b * a.kv0_bs + h * HD;
    const float* k1 = a.k1 + (size_t)b * a.kv1_bs + h * HD; const float* v1 = a.v1 + (size_t)b * a.kv1_bs + h * HD;

    for (int kb = 0; kb < nkb; ++kb) {
        const int kt = kb * KW + kw;
        const bool work = pair_ok && kt < nkt;                         // wave-uniform
        const int nact = min(KW, nkt - kb * KW);                       // partial tiles to sum
        const int k0n = kt * 32, k0s = min(k0n, Sk - 32);
        const AttSeg kg = att_seg_tile(k0, a.ld0, k1, a.ld1, a.S0, work ? k0s : 0);
        float kreg[HD / 2], vreg[HD / 2];
        f32x16 dk[NDT], dv[NDT];
        if (work) {
            att_load_row_form<HD>(kg, ql, half, kreg);
            att_load_row_form<HD>(att_seg_tile(v0, a.ld0, v1, a.ld1, a.S0, k0s), ql, half, vreg);
        }
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
        const bool keyok = k0s + ql >= k0n;                            // keys shared with the previous tile belong to that tile
        for (int u = 0; u < nqt; ++u) {
            const int u0n = u * 32, u0 = min(u0n, a.Sq - 32);
            // ---- stage Q, dO (LDS) and lse, D = rowsum(dO * O) of the 32 queries: C4 consecutive lanes own one row
            if (pair_ok) {
#pragma unroll
                for (int it = 0; it < ITS; ++it) {
                    if (it > 0 && (it & 1) == 0) __builtin_amdgcn_sched_barrier(0);       // at most two rows' worth of staging registers in flight
                    const int idx = tp + NT * it, row = idx / C4, c4 = idx % C4;
                    const float4 qv = *reinterpret_cast<const float4*>(qb + (size_t)(u0 + row) * a.ldq + 4 * c4);
                    const float4 gv = *reinterpret_cast<const float4*>(gb + (size_t)(u0 + row) * D + 4 * c4);
                    const float4 ov = *reinterpret_cast<const float4*>(ob + (size_t)(u0 + row) * D + 4 * c4);
                    *reinterpret_cast<float4*>(Qs + row * LDQ + 4 * c4) = qv;
                    *reinterpret_cast<float4*>(Gs + row * LDQ + 4 * c4) = gv;
                    float part = (gv.x * ov.x + gv.y * ov.y) + (gv.z * ov.z + gv.w * ov.w);
#pragma unroll
                    for (int m = C4 / 2; m >= 1; m >>= 1) part += __shfl_xor(part, m);
                    if (c4 == 0) { Ds[row] = part; Ls[row] = lb[u0 + row]; }
                }
            }
            __syncthreads();
            if (work) {
                f32x16 sa, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
                {
                    float fr[HD / 2];
                    const float* gp = Gs + ql * LDQ + half * (HD / 2);
#pragma unroll
                    for (int i = 0; i < HD / 8; ++i) { const float4 t = *reinterpret_cast<const float4*>(gp + 4 * i); fr[4 * i] = t.x; fr[4 * i + 1] = t.y; fr[4 * i + 2] = t.z; fr[4 * i + 3] = t.w; }
#pragma unroll
                    for (int s2 = 0; s2 < HD / 2; ++s2) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[s2], vreg[s2], dp, 0, 0, 0);
                    const float* qp = Qs + ql * LDQ + half * (HD / 2);
#pragma unroll
                    for (int i = 0; i < HD / 8; ++i) { const float4 t = *reinterpret_cast<const float4*>(qp + 4 * i); fr[4 * i] = t.x; fr[4 * i + 1] = t.y; fr[4 * i + 2] = t.z; fr[4 * i + 3] = t.w; }
#pragma unroll
                    for (int s2 = 0; s2 < HD / 2; ++s2) sa = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[s2], kreg[s2], sa, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // P and dS: register r <-> query u0 + f(r, half), lane <-> key k0s + ql
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 l4 = *reinterpret_cast<const float4*>(Ls + 8 * g + 4 * half);
                    const float4 d4 = *reinterpret_cast<const float4*>(Ds + 8 * g + 4 * half);
                    const float lr[4] = {l4.x, l4.y, l4.z, l4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * g + j;
                        const float pv = (keyok && u0 + 8 * g + 4 * half + j >= u0n) ? __expf(scale * sa[r] - lr[j]) : 0.f;
                        sa[r] = pv;
                        dp[r] = pv * (dp[r] - dr[j]) * scale;
                    }
                    // dS^t: T[key = ql][query = 8g + 4 half + 0..3]
                    *reinterpret_cast<float4*>(X + ql * LDT + 8 * g + 4 * half) = make_float4(dp[4 * g], dp[4 * g + 1], dp[4 * g + 2], dp[4 * g + 3]);
                }
                {   // dV^t += dO^t P
                    float fc[16][NDT];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float* rp = Gs + ATT_F(r, half) * LDQ + NDT * ql;
                        if constexpr (NDT == 2) { const float2 t = *reinterpret_cast<const float2*>(rp); fc[r][0] = t.x; fc[r][1] = t.y; } else fc[r][0] = rp[0];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int dt = 0; dt < NDT; ++dt) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fc[r][dt], sa[r], dv[dt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // dK^t += Q^t dS
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float* rp = Qs + ATT_F(r, half) * LDQ + NDT * ql;
                        if constexpr (NDT == 2) { const float2 t = *reinterpret_cast<const float2*>(rp); fc[r][0] = t.x; fc[r][1] = t.y; } else fc[r][0] = rp[0];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int dt = 0; dt < NDT; ++dt) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fc[r][dt], dp[r], dk[dt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // K of this wave's keys in col form (the A operand of the partial dQ; the only fragment that still comes from global memory: L2 hits, the
                // other wave of the SIMD owns the matrix pipe meanwhile)
                float kc[16][NDT];
                att_load_col_form<HD>(kg, ql, half, kc);
                // partial dQ^t = K^t dS^t over this wave's keys: dS^t[key = f(r, half)][query = ql] back from the wave's own tile
                float dst[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[r] = X[ATT_F(r, half) * LDT + ql];
                f32x16 dq[NDT];
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[r][dt], dst[r], dq[dt], 0, 0, 0);
                att_store_o<HD>(X + ql * LDQ, half, dq, 1.0f);         // (every dS^t value of this wave is in registers by now: same region)
            }
            __syncthreads();
            // ---- dQ rows of this query tile: sum of the pair's partial tiles in wave order (+ what earlier key blocks left in global memory)
            if (pair_ok) {
#pragma unroll
                for (int it = 0; it < ITS; ++it) {
                    const int idx = tp + NT * it, row = idx / C4, c4 = idx % C4;
                    float4 acc = *reinterpret_cast<const float4*>(Xp + row * LDQ + 4 * c4);
                    for (int w = 1; w < nact; ++w) {
                        const float4 t = *reinterpret_cast<const float4*>(Xp + (size_t)w * XSZ + row * LDQ + 4 * c4);
                        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
                    }
                    if (u0 + row >= u0n) {
                        float* dst4 = dqb + (size_t)(u0 + row) * a.ldq + 4 * c4;
                        if (kb > 0) {
                            const float4 t = *reinterpret_cast<const float4*>(dst4);
                            acc.x = t.x + acc.x; acc.y = t.y + acc.y; acc.z = t.z + acc.z; acc.w = t.w + acc.w;
                        }
                        *reinterpret_cast<float4*>(dst4) = acc;
                    }
                }
            }
            // (no barrier here: the next tile's staging writes the Q / dO region, which every wave finished reading before the barrier above; the
            //  partial region is written again only after the next staging barrier, by when every thread has left this reduction)
        }
        if (work) {
            const int key = k0s + ql;
            if (key >= k0n) {
                float* dkp = key < a.S0 ? a.dk0 + (size_t)b * a.kv0_bs + (size_t)key * a.ld0 + h * HD : a.dk1 + (size_t)b * a.kv1_bs + (size_t)(key - a.S0) * a.ld1 + h * HD;
                float* dvp = key < a.S0 ? a.dv0 + (size_t)b * a.kv0_bs + (size_t)key * a.ld0 + h * HD : a.dv1 + (size_t)b * a.kv1_bs + (size_t)(key - a.S0) * a.ld1 + h * HD;
                att_store_o<HD>(dkp, half, dk, 1.0f);
                att_store_o<HD>(dvp, half, dv, 1.0f);
            }
        }
    }
}

static const bool g_attn_small = [] { const char* e = getenv("ACT_ATTN_SMALL"); return !(e && e[0] == '0'); }();      // dev A/B knob

static const bool g_attn_reg = [] { const char* e = getenv("ACT_ATTN_REG"); return !(e && e[0] == '0'); }();          // dev A/B knob: 0 = LDS-staged kernels
// the register-resident kernels want full 32-row tiles inside one key segment (see the tail rule above)
static inline bool attn_reg_ok(int Sq, int S0, int S1) { return g_attn_reg && Sq >= 32 && S1 >= 32 && (S0 % 32) == 0; }
// forward: the LDS-staged kernel (K / V of a pair shared by the waves of a workgroup) is the faster one on 8 of 10 measured shapes; the
// register-resident forward stays selectable for A/B runs (ACT_ATTN_FWD_REG=1)
static const bool g_attn_fwd_reg = [] { const char* e = getenv("ACT_ATTN_FWD_REG"); return e && e[0] == '1'; }();

static int launch_attn_fwd_reg(const AttnRegArgs& a, int head_dim, hipStream_t s) {
    const long long items = (long long)a.B * a.H * ((a.Sq + 31) / 32);
    const unsigned grid = (unsigned)((items + 3) / 4);
    if (head_dim == 64) hipLaunchKernelGGL(attn_fwd_reg_kernel<64>, dim3(grid), dim3(256), 0, s, a);
    else                hipLaunchKernelGGL(attn_fwd_reg_kernel<32>, dim3(grid), dim3(256), 0, s, a);
    ACT_LAUNCH_CHECK();
    return 0;
}
// ACT_ATTN_BWD_ONE: 1 (default) = the single-pass kernel wherever the register kernels' tile conditions hold, 0 = the two-role kernel (A/B)
static const bool g_attn_bwd_one = [] { const char* e = getenv("ACT_ATTN_BWD_ONE"); return !(e && e[0] == '0'); }();
template <int HD, int KW, int NP>
static int launch_attn_bwd_one_t(const AttnRegArgs& a, hipStream_t s) {
    const size_t smem = ((size_t)NP * (2 * 32 * (HD + 4) + 64) + (size_t)NP * KW * 32 * (HD + 4)) * sizeof(float);
    auto k = attn_bwd_one_kernel<HD, KW, NP>;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    const long long pairs = (long long)a.B * a.H;
    hipLaunchKernelGGL(k, dim3((unsigned)((pairs + NP - 1) / NP)), dim3(64 * KW * NP), smem, s, a);
    ACT_LAUNCH_CHECK();
    return 0;
}
static int launch_attn_bwd_one(const AttnRegArgs& a, int head_dim, hipStream_t s) {
    const int nkt = (a.S0 + a.S1 + 31) / 32;
    // pairs per workgroup at KW = 2 (ACT_ATTN_BWD_NP = 1 | 2): one pair per 128-thread workgroup spreads the 768 pairs of the student decoder over three workgroups
    // per CU -- 50.3 -> 48.0 us (profiles/r06_attn_bwd_np_ab.txt)
    static const int np2 = [] { const char* e = getenv("ACT_ATTN_BWD_NP"); return e ? atoi(e) : 1; }();
    if (head_dim == 64) return nkt <= 2 ? (np2 == 1 ? launch_attn_bwd_one_t<64, 2, 1>(a, s) : launch_attn_bwd_one_t<64, 2, 2>(a, s)) : launch_attn_bwd_one_t<64, 4, 1>(a, s);
    return nkt <= 2 ? (np2 == 1 ? launch_attn_bwd_one_t<32, 2, 1>(a, s) : launch_attn_bwd_one_t<32, 2, 2>(a, s)) : launch_attn_bwd_one_t<32, 4, 1>(a, s);
}
static int launch_attn_bwd_reg(const AttnRegArgs& a, int head_dim, hipStream_t s) {
    if (g_attn_bwd_one) return launch_attn_bwd_one(a, head_dim, s);
    const long long pairs = (long long)a.B * a.H;
    const unsigned ndq = (unsigned)((pairs * ((a.Sq + 31) / 32) + 3) / 4), ndkv = (unsigned)((pairs * ((a.S0 + a.S1 + 31) / 32) + 3) / 4);
    // the heavier dK / dV items (128 MFMAs per tile pair) are dispatched first? no: dQ blocks first -- their stores are the ones a following
    // GEMM (dn1 = dqkv . W) waits for in full anyway; the order only shapes the tail
    if (head_dim == 64) hipLaunchKernelGGL(attn_bwd_reg_kernel<64>, dim3(ndq + ndkv), dim3(256), 0, s, a, (int)ndq);
    else                hipLaunchKernelGGL(attn_bwd_reg_kernel<32>, dim3(ndq + ndkv), dim3(256), 0, s, a, (int)ndq);
    ACT_LAUNCH_CHECK();
    return 0;
}

template <int HD>
static int launch_attn_bwd_mfma_t(const AttnBwdArgs& a, hipStream_t s) {
    const size_t smem = ((size_t)3 * 64 * HD + 64 * 65 + 128) * sizeof(float);
    static const int occ = [] { const char* e = getenv("ACT_ATTN_BWD_OCC"); return e ? atoi(e) : 2; }();     // dev knob
    auto k = occ == 1 ? attn_bwd_mfma_kernel<HD, 1> : attn_bwd_mfma_kernel<HD, 2>;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)(a.B * a.H)), dim3(256), smem, s, a);
    ACT_LAUNCH_CHECK();
    return 0;
}
static int launch_attn_bwd_mfma(const AttnBwdArgs& a, int head_dim, hipStream_t s) {
    return head_dim == 64 ? launch_attn_bwd_mfma_t<64>(a, s) : launch_attn_bwd_mfma_t<32>(a, s);
}

// dev A/B knob, default OFF: 1 = V staged transposed (kernel comment).  Measured 5-7 % SLOWER on every shape (teacher prefix 64 q x 128 k 45.1 -> 48.3 us,
// S = 512 226 -> 238 us, profiles/r05_attn_vt_ab.txt): the four dword loads per quad cost more than the sixteen exposed ds_read_b32 round trips they remove --
// three waves per SIMD already hide those.  Kept as a measured non-improvement; bit-identical either way.
static const bool g_attn_vt = [] { const char* e = getenv("ACT_ATTN_VT"); return e && e[0] == '1'; }();
// dev A/B knobs (round 6): ACT_ATTN_FWD_NW=2 -> one pair per 128-thread workgroup where QT == 2; ACT_ATTN_FWD_PRIO=1 -> s_setprio around the MFMA bursts
static const int g_attn_fwd_nw = [] { const char* e = getenv("ACT_ATTN_FWD_NW"); return e ? atoi(e) : 4; }();
static const bool g_attn_fwd_prio = [] { const char* e = getenv("ACT_ATTN_FWD_PRIO"); return e && e[0] == '1'; }();
template <int HD, int JT, int QT, bool VT, int NW, bool PRIO, int DIAG = 0>
static int launch_attn_fwd4(const AttnFwdArgs& a0, hipStream_t s) {
    constexpr int pairs = NW / QT;
    static const int stg = [] { const char* e = getenv("ACT_ATTN_FWD_STAGGER"); return e ? atoi(e) : 0; }();          // dev A/B knob (kernel comment)
    static const int stg_mod = [] { const char* e = getenv("ACT_ATTN_FWD_STAGGER_MOD"); return e ? atoi(e) : 3; }();
    static const int sprio = [] { const char* e = getenv("ACT_ATTN_FWD_SLOT_PRIO"); return e ? atoi(e) : 0; }();   // dev A/B knob (attn_slot_prio)
    AttnFwdArgs a = a0; a.slot_prio = sprio; a.stagger = stg; a.stagger_mod = stg_mod > 0 ? stg_mod : 1;
    const size_t smem = (size_t)pairs * (JT * 32 * (HD + 4) + (VT ? HD * (JT * 32 + 4) : JT * 32 * (HD + 4))) * sizeof(float);
    const long long np = (long long)a.B * a.H;
    const unsigned gx = (unsigned)((np + pairs - 1) / pairs), gy = (unsigned)((a.Sq + QT * 32 - 1) / (QT * 32));
    auto k = attn_fwd_kernel<HD, JT, QT, VT, NW, PRIO, DIAG>;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k, dim3(gx, gy), dim3(NW * 64), smem, s, a);
    ACT_LAUNCH_CHECK();
    return 0;
}
template <int HD, int JT, int QT>
static int launch_attn_fwd3(const AttnFwdArgs& a, hipStream_t s) {
    const bool vt = g_attn_vt && (a.S0 % (JT * 32)) == 0;              // a key chunk must not straddle the two key segments (kernel comment)
#ifdef ACT_ATTN_DIAG                                                     // only in a library built with ACT_HIPCC_EXTRA=-DACT_ATTN_DIAG (benchmarks/scripts/r06_run31.sh)
    if constexpr (QT == 2 && JT == 1 && HD == 64) {                     // dev ablation of the teacher shape (kernel comment; wrong results)
        static const int diag = [] { const char* e = getenv("ACT_ATTN_FWD_DIAG"); return e ? atoi(e) : 0; }();
#define DG(D) if (diag == D) return launch_attn_fwd4<HD, JT, QT, false, 4, false, D>(a, s)
        DG(1); DG(2); DG(3); DG(4); DG(8); DG(16); DG(24); DG(28); DG(31); DG(7); DG(32); DG(35); DG(39); DG(56); DG(63);
#undef DG
    }
#endif
    if constexpr (QT == 2 && JT <= 2) {
        if (!vt && g_attn_fwd_nw == 2) return g_attn_fwd_prio ? launch_attn_fwd4<HD, JT, QT, false, 2, true>(a, s) : launch_attn_fwd4<HD, JT, QT, false, 2, false>(a, s);
        if (!vt && g_attn_fwd_prio) return launch_attn_fwd4<HD, JT, QT, false, 4, true>(a, s);
    }
    return vt ? launch_attn_fwd4<HD, JT, QT, true, 4, false>(a, s) : launch_attn_fwd4<HD, JT, QT, false, 4, false>(a, s);
}
template <int HD>
static int launch_attn_fwd(const AttnFwdArgs& a, hipStream_t s) {
    const int Sk = a.S0 + a.S1;
    int JT = Sk > 128 ? 4 : (Sk + 31) / 32;                          // key tiles per LDS chunk
    const int QT = a.Sq > 128 ? 4 : (a.Sq + 31) / 32;                // query tiles per workgroup
    // K / V of all pairs of a workgroup live in LDS: beyond ~80 KB only one workgroup fits a CU and the kernel turns into pure latency.
    // 64-key chunks with the online softmax halve the footprint (teacher prompt-prefix shape 64 q x 128 k: 139 -> 70 KB, 70 -> 57 us).
    const int pairs = QT == 1 ? 4 : (QT == 2 ? 2 : 1);
    if (JT == 4 && QT <= 2 && (size_t)pairs * 2 * JT * 32 * (HD + 4) * sizeof(float) > 80 * 1024) JT = 2;
    // round 3: from 128 keys on, ONE 32-key tile per LDS chunk -- 35 KB instead of 70 KB per workgroup = four instead of two workgroups per CU, whose
    // staging / softmax / MFMA phases then overlap (Stage-I prefix shape 56.9 -> 51.8 us, S = 128 83 -> 76 us, stress teacher 64 + 512 320 -> 283 us;
    // shorter sequences keep the single chunk: 64 keys 20.6 vs 21.9 us).  ACT_ATTN_JT = 1..4 forces a chunk size (A/B)
    static const int jt_env = [] { const char* e = getenv("ACT_ATTN_JT"); return e ? atoi(e) : 0; }();
    if (jt_env >= 1 && jt_env <= 4) { if (jt_env < JT && Sk > jt_env * 32) JT = jt_env; }
    else if (Sk >= 128) JT = 1;
#define C3(J, Q) if (JT == J && QT == Q) return launch_attn_fwd3<HD, J, Q>(a, s)
    C3(1, 1); C3(1, 2); C3(1, 3); C3(1, 4); C3(2, 1); C3(2, 2); C3(2, 3); C3(2, 4); C3(3, 1); C3(3, 2); C3(3, 3); C3(3, 4); C3(4, 1); C3(4, 2); C3(4, 3); C3(4, 4);
#undef C3
    return ACT_E_BADARG;
}

extern "C" int act_attention_fwd_f32(const float* qkv, float* out, float* lse, int B, int S, int H, int head_dim, float scale,
                                     act_stream_t stream) {
    if (!qkv || !out) return ACT_E_NULLPTR;
    if (B < 0 || S <= 0 || H <= 0 || (head_dim != 64 && head_dim != 32)) return ACT_E_BADARG;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_ATTN_FWD, s, 4.0 * B * H * (double)S * S * head_dim, 16.0 * B * S * (double)H * head_dim);
    const int D = H * head_dim;
    if (S <= 16 && g_attn_small) {                                     // register-resident kernel: one wave per (cloud, head)
        const unsigned grid = (unsigned)(((long long)B * H + 3) / 4);
        if (head_dim == 64) hipLaunchKernelGGL(attn_small_fwd_kernel<64>, dim3(grid), dim3(256), 0, s, qkv, out, lse, B, S, H, scale);
        else                hipLaunchKernelGGL(attn_small_fwd_kernel<32>, dim3(grid), dim3(256), 0, s, qkv, out, lse, B, S, H, scale);
        ACT_LAUNCH_CHECK();
        return 0;
    }
    if (g_attn_fwd_reg && attn_reg_ok(S, 0, S)) {
        AttnRegArgs r{};
        r.q = qkv; r.k0 = qkv; r.v0 = qkv; r.k1 = qkv + D; r.v1 = qkv + 2 * D;
        r.q_bs = (long long)S * 3 * D; r.kv0_bs = 0; r.kv1_bs = r.q_bs; r.ldq = 3 * D; r.ld0 = 0; r.ld1 = 3 * D;
        r.B = B; r.H = H; r.Sq = S; r.S0 = 0; r.S1 = S; r.scale = scale; r.o = out; r.lse_out = lse;
        return launch_attn_fwd_reg(r, head_dim, s);
    }
    AttnFwdArgs a{};
    a.q = qkv; a.k0 = nullptr; a.v0 = nullptr; a.k1 = qkv + D; a.v1 = qkv + 2 * D;
    a.q_bs = (long long)S * 3 * D; a.kv0_bs = 0; a.kv1_bs = a.q_bs; a.ldq = 3 * D; a.ld0 = 0; a.ld1 = 3 * D;
    a.B = B; a.H = H; a.Sq = S; a.S0 = 0; a.S1 = S; a.scale = scale; a.out = out; a.lse = lse;
    return head_dim == 64 ? launch_attn_fwd<64>(a, s) : launch_attn_fwd<32>(a, s);
}

// queries: the Sq rows of qkv1 [B,Sq,3,H,hd]; keys/values: S0 rows of kv0 [B,S0,2,H,hd] followed by the Sq rows of qkv1.
// (teacher ViT of models/dvae.py:536-576: prompt tokens only ever act as keys/values -- their outputs are replaced by the
//  next layer's prompts -- so queries, projection and MLP are evaluated for the patch tokens only.)
extern "C" int act_attention_fwd_prefix_f32(const float* kv0, int S0, const float* qkv1, int Sq, float* out, float* lse, int B,
                                            int H, int head_dim, float scale, act_stream_t stream) {
    if (!kv0 || !qkv1 || !out) return ACT_E_NULLPTR;
    if (B < 0 || Sq <= 0 || S0 < 0 || H <= 0 || (head_dim != 64 && head_dim != 32)) return ACT_E_BADARG;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_ATTN_FWD, s, 4.0 * B * H * (double)Sq * (S0 + Sq) * head_dim, 4.0 * B * (double)H * head_dim * (4.0 * Sq + 2.0 * S0));
    const int D = H * head_dim;
    if (g_attn_fwd_reg && attn_reg_ok(Sq, S0, Sq)) {
        AttnRegArgs r{};
        r.q = qkv1; r.k0 = kv0; r.v0 = kv0 + D; r.k1 = qkv1 + D; r.v1 = qkv1 + 2 * D;
        r.q_bs = (long long)Sq * 3 * D; r.kv0_bs = (long long)S0 * 2 * D; r.kv1_bs = r.q_bs; r.ldq = 3 * D; r.ld0 = 2 * D; r.ld1 = 3 * D;
        r.B = B; r.H = H; r.Sq = Sq; r.S0 = S0; r.S1 = Sq; r.scale = scale; r.o = out; r.lse_out = lse;
        return launch_attn_fwd_reg(r, head_dim, s);
    }
    AttnFwdArgs a{};
    a.q = qkv1; a.k0 = kv0; a.v0 = kv0 + D; a.k1 = qkv1 + D; a.v1 = qkv1 + 2 * D;
    a.q_bs = (long long)Sq * 3 * D; a.kv0_bs = (long long)S0 * 2 * D; a.kv1_bs = a.q_bs; a.ldq = 3 * D; a.ld0 = 2 * D; a.ld1 = 3 * D;
    a.B = B; a.H = H; a.Sq = Sq; a.S0 = S0; a.S1 = Sq; a.scale = scale; a.out = out; a.lse = lse;
    return head_dim == 64 ? launch_attn_fwd<64>(a, s) : launch_attn_fwd<32>(a, s);
}

// the same forward with the output ALSO (out != NULL) or ONLY (out == NULL) as (hi, lo) bf16 planes [B*Sq][H*hd]: the A operand of a split-bf16 projection
// (opt-in teacher path, gemm_bf16x3.hip).  LDS-staged kernel only.
extern "C" int act_attention_fwd_prefix_planes_f32(const float* kv0, int S0, const float* qkv1, int Sq, float* out, uint16_t* out_hi, uint16_t* out_lo,
                                                   float* lse, int B, int H, int head_dim, float scale, act_stream_t stream) {
    if (!kv0 || !qkv1 || !out_hi || !out_lo) return ACT_E_NULLPTR;
    if (B < 0 || Sq <= 0 || S0 < 0 || H <= 0 || (head_dim != 64 && head_dim != 32) || (((uintptr_t)out_hi | (uintptr_t)out_lo) & 7)) return ACT_E_BADARG;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_ATTN_FWD, s, 4.0 * B * H * (double)Sq * (S0 + Sq) * head_dim, 4.0 * B * (double)H * head_dim * (4.0 * Sq + 2.0 * S0));
    const int D = H * head_dim;
    AttnFwdArgs a{};
    a.q = qkv1; a.k0 = kv0; a.v0 = kv0 + D; a.k1 = qkv1 + D; a.v1 = qkv1 + 2 * D;
    a.q_bs = (long long)Sq * 3 * D; a.kv0_bs = (long long)S0 * 2 * D; a.kv1_bs = a.q_bs; a.ldq = 3 * D; a.ld0 = 2 * D; a.ld1 = 3 * D;
    a.B = B; a.H = H; a.Sq = Sq; a.S0 = S0; a.S1 = Sq; a.scale = scale; a.out = out; a.lse = lse; a.out_hi = out_hi; a.out_lo = out_lo;
    return head_dim == 64 ? launch_attn_fwd<64>(a, s) : launch_attn_fwd<32>(a, s);
}

extern "C" int act_attention_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                                     int B, int S, int H, int head_dim, float scale, act_stream_t stream) {
    if (!qkv || !out || !dout || !lse || !dqkv) return ACT_E_NULLPTR;
    if (B < 0 || S <= 0 || H <= 0 || (head_dim != 64 && head_dim != 32)) return ACT_E_BADARG;
    if (B == 0) return 0;
    static const bool use_valu = [] { const char* e = getenv("ACT_ATTN_BWD_VALU"); return e && e[0] == '1'; }();   // dev A/B knob
    if (S <= 16 && g_attn_small) {
        hipStream_t s = (hipStream_t)stream;
        ActProfScope ps(KID_ATTN_BWD, s, 10.0 * B * H * (double)S * S * head_dim, 28.0 * B * S * (double)H * head_dim);
        const unsigned grid = (unsigned)(((long long)B * H + 3) / 4);
        if (head_dim == 64) hipLaunchKernelGGL(attn_small_bwd_kernel<64>, dim3(grid), dim3(256), 0, s, qkv, out, dout, lse, dqkv, B, S, H, scale);
        else                hipLaunchKernelGGL(attn_small_bwd_kernel<32>, dim3(grid), dim3(256), 0, s, qkv, out, dout, lse, dqkv, B, S, H, scale);
        ACT_LAUNCH_CHECK();
        return 0;
    }
    if (!use_valu && attn_reg_ok(S, 0, S)) {
        hipStream_t s = (hipStream_t)stream;
        ActProfScope ps(KID_ATTN_BWD, s, 10.0 * B * H * (double)S * S * head_dim, 28.0 * B * S * (double)H * head_dim);
        const int D = H * head_dim;
        AttnRegArgs r{};
        r.q = qkv; r.k0 = qkv; r.v0 = qkv; r.k1 = qkv + D; r.v1 = qkv + 2 * D; r.out = out; r.dout = dout; r.lse = lse;
        r.dq = dqkv; r.dk0 = dqkv; r.dv0 = dqkv; r.dk1 = dqkv + D; r.dv1 = dqkv + 2 * D;
        r.q_bs = (long long)S * 3 * D; r.kv0_bs = 0; r.kv1_bs = r.q_bs; r.ldq = 3 * D; r.ld0 = 0; r.ld1 = 3 * D;
        r.B = B; r.H = H; r.Sq = S; r.S0 = 0; r.S1 = S; r.scale = scale;
        return launch_attn_bwd_reg(r, head_dim, s);
    }
    if (!use_valu) {
        hipStream_t s = (hipStream_t)stream;
        ActProfScope ps(KID_ATTN_BWD, s, 10.0 * B * H * (double)S * S * head_dim, 28.0 * B * S * (double)H * head_dim);
        AttnBwdArgs a;
        a.kv0 = nullptr; a.qkv1 = qkv; a.out = out; a.dout = dout; a.lse = lse; a.dkv0 = nullptr; a.dqkv1 = dqkv;
        a.B = B; a.S0 = 0; a.Sq = S; a.H = H; a.scale = scale;
        return launch_attn_bwd_mfma(a, head_dim, s);
    }
    const int S4 = (S + 3) & ~3;
    const int KR = S4 < ATT_KC ? S4 : ATT_KC;
    const size_t smem = ((size_t)2 * KR * (head_dim + 4) + (size_t)2 * ATT_QC * (head_dim + 4) + (size_t)ATT_QC * (KR + 4) + 2 * ATT_QC) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_ATTN_BWD, s, 10.0 * B * H * (double)S * S * head_dim, 28.0 * B * S * (double)H * head_dim);
#define BWD(HD) { auto k = attn_bwd_kernel<HD>; \
        if (smem > 48 * 1024) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); if (e != hipSuccess) return (int)e; } \
        hipLaunchKernelGGL(k, dim3((unsigned)(B * H)), dim3(256), smem, s, qkv, out, dout, lse, dqkv, B, S, H, scale, KR); }
    if (head_dim == 64) BWD(64) else BWD(32)
#undef BWD
    ACT_LAUNCH_CHECK();
    return 0;
}

// backward of act_attention_fwd_prefix_f32: dqkv1 [B,Sq,3,H,hd] (dQ, and dK/dV of the Sq own rows), dkv0 [B,S0,2,H,hd].
extern "C" int act_attention_bwd_prefix_f32(const float* kv0, int S0, const float* qkv1, int Sq, const float* out, const float* dout,
                                            const float* lse, float* dkv0, float* dqkv1, int B, int H, int head_dim, float scale,
                                            act_stream_t stream) {
    if (!qkv1 || !out || !dout || !lse || !dqkv1 || (S0 > 0 && (!kv0 || !dkv0))) return ACT_E_NULLPTR;
    if (B < 0 || Sq <= 0 || S0 < 0 || H <= 0 || (head_dim != 64 && head_dim != 32)) return ACT_E_BADARG;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_ATTN_BWD, s, 10.0 * B * H * (double)Sq * (S0 + Sq) * head_dim,
                    4.0 * B * (double)H * head_dim * (8.0 * Sq + 4.0 * S0));
    if (attn_reg_ok(Sq, S0, Sq)) {
        const int D = H * head_dim;
        AttnRegArgs r{};
        r.q = qkv1; r.k0 = S0 > 0 ? kv0 : qkv1; r.v0 = S0 > 0 ? kv0 + D : qkv1; r.k1 = qkv1 + D; r.v1 = qkv1 + 2 * D;
        r.out = out; r.dout = dout; r.lse = lse;
        r.dq = dqkv1; r.dk0 = S0 > 0 ? dkv0 : dqkv1; r.dv0 = S0 > 0 ? dkv0 + D : dqkv1; r.dk1 = dqkv1 + D; r.dv1 = dqkv1 + 2 * D;
        r.q_bs = (long long)Sq * 3 * D; r.kv0_bs = (long long)S0 * 2 * D; r.kv1_bs = r.q_bs; r.ldq = 3 * D; r.ld0 = 2 * D; r.ld1 = 3 * D;
        r.B = B; r.H = H; r.Sq = Sq; r.S0 = S0; r.S1 = Sq; r.scale = scale;
        return launch_attn_bwd_reg(r, head_dim, s);
    }
    AttnBwdArgs a;
    a.kv0 = kv0; a.qkv1 = qkv1; a.out = out; a.dout = dout; a.lse = lse; a.dkv0 = dkv0; a.dqkv1 = dqkv1;
    a.B = B; a.S0 = S0; a.Sq = Sq; a.H = H; a.scale = scale;
    return launch_attn_bwd_mfma(a, head_dim, s);
}
