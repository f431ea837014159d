// composite.hip -- host-side launch sequences: one C call enqueues every kernel of a module (include/act_hip.h, "composite
// entry points").  No device code here: each function is the fixed launch schedule of one module of the path, built from the
// single-kernel entry points of this library, so results are bit-identical to issuing those calls one by one from the host
// language -- what changes is the host cost (one FFI crossing + ~4 us per hipLaunchKernel instead of ~35 us of Python per launch).
//
// Stream discipline: everything is enqueued on `stream`; weight-gradient GEMMs + bias column sums of a block backward go to
// `side_stream` when one is given: fork = event recorded on `stream` after the producing kernel, join = event recorded on
// `side_stream` that `stream` waits for before the function returns.  Buffers handed to the side stream are caller-owned slabs
// that outlive the call on `stream` order, so the caller's allocator never sees a cross-stream hazard.
#include "common.h"
#include <mutex>
#include <unordered_map>
#include <vector>
#include <stdlib.h>

#define CK(expr) do { const int rc__ = (expr); if (rc__ != 0) return rc__; } while (0)
// every non-GEMM launch goes through RUN: skipped while the calling thread collects the GEMM shapes of a composite (see below)
#define RUN(expr) do { if (!t_collect) { const int rc__ = (expr); if (rc__ != 0) return rc__; } } while (0)

namespace {

// ---- shape collection: between act_composite_collect_begin / _end (same host thread) the composite entry points launch
// nothing and only record the (a_kmajor, b_kmajor, M, N, K) of every GEMM they would launch, so the host-side autotuner can
// time and register configurations for them BEFORE their first real execution (results then never depend on call history)
struct GemmShape { int ak, bk, M, N, K; };
thread_local std::vector<GemmShape>* t_collect = nullptr;
inline bool collecting(int ak, int bk, int M, int N, int K) {
    if (!t_collect) return false;
    t_collect->push_back({ak, bk, M, N, K});
    return true;
}

// ---- fork / join events: a small ring per side stream (an event may be re-recorded while an earlier wait on it is pending:
// hipStreamWaitEvent captures the record that is current at call time; the ring only keeps the number of live records small)
struct EventRing {
    std::vector<hipEvent_t> ev;
    size_t next = 0;
    hipEvent_t get() {
        if (ev.size() < 32) { hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming); ev.push_back(e); return e; }
        hipEvent_t e = ev[next]; next = (next + 1) % ev.size(); return e;
    }
};
std::mutex g_ring_mu;
std::unordered_map<hipStream_t, EventRing> g_rings;

int order_after(hipStream_t waiter, hipStream_t producer) {            // work enqueued later on `waiter` runs after everything on `producer`
    if (t_collect) return 0;
    hipEvent_t e;
    { std::lock_guard<std::mutex> g(g_ring_mu); e = g_rings[producer].get(); }
    hipError_t r = hipEventRecord(e, producer);
    if (r != hipSuccess) return (int)r;
    r = hipStreamWaitEvent(waiter, e, 0);
    return (int)r;
}

inline float attn_scale(int hd) { return (float)pow((double)hd, -0.5); }      // == Python's float(hd) ** -0.5
inline act_gemm_epilogue_t epi0() { act_gemm_epilogue_t e{}; e.alpha = 1.0f; return e; }

// C[M,N] = epi(A[M,K] . W[N,K]^T)            forward Linear
int gemm_nt(int M, int N, int K, const float* A, int lda, const float* W, int ldw, float* C, int ldc, const act_gemm_epilogue_t& e,
            float* ws, size_t wsb, hipStream_t s) {
    if (collecting(1, 1, M, N, K)) return 0;
    return act_sgemm_f32(1, 1, M, N, K, A, lda, W, ldw, C, ldc, &e, ws, wsb, s);
}
// dX[M,K'] = epi(dY[M,N'] . W[N',K'])        input gradient: A = dY [M][N'] K-major, B = W stored [N'][K'] = [K][N] N-major
int gemm_nn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, const act_gemm_epilogue_t& e,
            float* ws, size_t wsb, hipStream_t s) {
    if (collecting(1, 0, M, N, K)) return 0;
    return act_sgemm_f32(1, 0, M, N, K, A, lda, B, ldb, C, ldc, &e, ws, wsb, s);
}
// dW[M,N] = dY[K,M]^T . X[K,N]               weight gradient: both operands stored [K][*]
int gemm_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, float* ws, size_t wsb, hipStream_t s) {
    if (collecting(0, 0, M, N, K)) return 0;
    const act_gemm_epilogue_t e = epi0();
    return act_sgemm_f32(0, 0, M, N, K, A, lda, B, ldb, C, ldc, &e, ws, wsb, s);
}
int colsum(const float* in, int R, int C, float* out, float* ws, size_t wsb, hipStream_t s) {
    if (t_collect) return 0;
    if (wsb < act_colsum_workspace(R, C)) return ACT_E_BADARG;
    return act_colsum_f32(in, R, C, C, out, 0, ws, wsb, s);
}

struct Carver {                      // hands out consecutive, 4-float aligned pieces of a slab
    float* p; size_t used = 0;
    explicit Carver(float* base) : p(base) {}
    float* take(size_t n) { float* r = p ? p + used : nullptr; used += (n + 3) & ~(size_t)3; return r; }
};

// ---- Transformer block ---------------------------------------------------------------------------------------------------
struct BlockSaved { float *xin, *n1, *qkv, *att, *x1, *n2, *hpre, *a, *mean1, *rstd1, *mean2, *rstd2, *lse; };
size_t carve_block(float* base, const act_block_dims_t& d, BlockSaved& sv) {
    const size_t T = (size_t)d.B * d.S, D = d.D, Hd = d.hidden;
    Carver c(base);
    sv.xin = c.take(T * D); sv.n1 = c.take(T * D); sv.qkv = c.take(T * 3 * D); sv.att = c.take(T * D); sv.x1 = c.take(T * D);
    sv.n2 = c.take(T * D); sv.hpre = c.take(T * Hd); sv.a = c.take(T * Hd);
    sv.mean1 = c.take(T); sv.rstd1 = c.take(T); sv.mean2 = c.take(T); sv.rstd2 = c.take(T); sv.lse = c.take((size_t)d.B * d.heads * d.S);
    return c.used;
}
struct BlockBwdScratch { float *dy2, *dh, *dn2, *dx1, *dy1, *datt, *dqkv, *dn1; };
size_t carve_block_bwd(float* base, const act_block_dims_t& d, BlockBwdScratch& sc) {
    const size_t T = (size_t)d.B * d.S, D = d.D, Hd = d.hidden;
    Carver c(base);
    sc.dy2 = c.take(T * D); sc.dh = c.take(T * Hd); sc.dn2 = c.take(T * D); sc.dx1 = c.take(T * D); sc.dy1 = c.take(T * D);
    sc.datt = c.take(T * D); sc.dqkv = c.take(T * 3 * D); sc.dn1 = c.take(T * D);
    return c.used;
}
bool bad_dims(const act_block_dims_t* d) {
    return !d || d->B <= 0 || d->S <= 0 || d->D <= 0 || d->heads <= 0 || d->D % d->heads || d->hidden <= 0;
}

// ---- prefix block --------------------------------------------------------------------------------------------------------
struct PrefixSaved { float *n1p, *meanp, *rstdp, *kvp, *xin, *n1x, *mean1, *rstd1, *qkvx, *att, *lse, *x1, *n2, *mean2, *rstd2, *hpre, *a; };
size_t carve_prefix(float* base, const act_block_dims_t& d, int P, PrefixSaved& sv) {
    const size_t TG = (size_t)d.B * d.S, TP = (size_t)d.B * P, D = d.D, Hd = d.hidden;
    Carver c(base);
    sv.n1p = c.take(TP * D); sv.kvp = c.take(TP * 2 * D); sv.xin = c.take(TG * D); sv.n1x = c.take(TG * D); sv.qkvx = c.take(TG * 3 * D);
    sv.att = c.take(TG * D); sv.x1 = c.take(TG * D); sv.n2 = c.take(TG * D); sv.hpre = c.take(TG * Hd); sv.a = c.take(TG * Hd);
    sv.meanp = c.take(TP); sv.rstdp = c.take(TP); sv.mean1 = c.take(TG); sv.rstd1 = c.take(TG); sv.mean2 = c.take(TG); sv.rstd2 = c.take(TG);
    sv.lse = c.take((size_t)d.B * d.heads * d.S);
    return c.used;
}
struct PrefixBwdScratch { float *dh, *dn2, *dx1, *datt, *dkvp, *dqkvx, *dn1x, *dn1p; };
size_t carve_prefix_bwd(float* base, const act_block_dims_t& d, int P, PrefixBwdScratch& sc) {
    const size_t TG = (size_t)d.B * d.S, TP = (size_t)d.B * P, D = d.D, Hd = d.hidden;
    Carver c(base);
    sc.dh = c.take(TG * Hd); sc.dn2 = c.take(TG * D); sc.dx1 = c.take(TG * D); sc.datt = c.take(TG * D); sc.dkvp = c.take(TP * 2 * D);
    sc.dqkvx = c.take(TG * 3 * D); sc.dn1x = c.take(TG * D); sc.dn1p = c.take(TP * D);
    return c.used;
}

// the launch sequence of one prefix block given the LayerNorm'd prompt rows n1p (shared by the inference stack and the
// differentiable forward)
// OPT-IN (act_prefix_vit_fwd_bf16x3_f32): the bf16 (hi, lo) planes of one block's four weights + the scratch its activations are split into
struct X3Block { const uint16_t *qkv, *proj, *fc1, *fc2; uint16_t* a_planes; size_t a_elems; bool n1p_is_planes; };
// scratch layout of the split-bf16 path: [0, 2 TG Hd) the planes of the MLP's hidden activation (written by fc1's epilogue, read by fc2), then ONE region of
// 2 max(TG, TP) D for the planes of whichever other activation is multiplied next (prompt rows -> n1x -> attention output -> n2: each is consumed by the
// product that follows it on the stream before the next producer overwrites it)
inline bool x3_fits(const act_block_dims_t& d, int P, size_t a_elems) {
    const size_t TG = (size_t)d.B * d.S, TP = (size_t)d.B * P;
    return a_elems >= 2 * TG * d.hidden + 2 * (TG > TP ? TG : TP) * d.D;
}

int prefix_block_core(const act_block_dims_t& d, int P, const act_block_params_t& w, const float* x, const float* pos, const float* n1p,
                      bool keep, PrefixSaved& sv, float* out, float* ws, size_t wsb, hipStream_t s, const X3Block* x3 = nullptr) {
    const int B = d.B, G = d.S, D = d.D, H = d.heads, hd = D / H, Hd = d.hidden, TG = B * G, TP = B * P;
    // C = epi(A . W^T): on the f32-input MFMA kernels -- or, for the frozen teacher with the split-bf16 switch on and a shape the kernel takes, A is split into
    // (hi, lo) bf16 planes and multiplied with the weight's planes (W_hi = the sub-block of the plane image; lo plane `wplane` elements behind it)
    uint16_t* hid_planes = x3 ? x3->a_planes : nullptr;
    uint16_t* tmp_planes = x3 ? x3->a_planes + (size_t)2 * TG * Hd : nullptr;
    // (keep: the differentiable forward of Stage-I prompt tuning.  Its backward never needs n1x / n2 / a in fp32 -- the block weights are frozen, no dW --, only the
    //  LayerNorm statistics, the pre-GELU values, xin, x1, qkvx, kvp, att and lse, all of which are still written)
    const bool x3_ok = x3 && x3_fits(d, P, x3->a_elems);
    // `pre`: A is already there as planes (hi at pre, lo M*K further); `emit`: write the result as planes there instead of fp32 C
    auto linear = [&](int M, int N, int K, const float* A, const float* W, const uint16_t* W_hi, size_t wplane, float* C, const act_gemm_epilogue_t& e,
                      const uint16_t* pre = nullptr, uint16_t* emit = nullptr) -> int {
        if (x3_ok && W_hi && act_sgemm_nt_bf16x3_supported(M, N, K)) {
            if (t_collect) return 0;
            const uint16_t* ah = pre; const uint16_t* al = pre ? pre + (size_t)M * K : nullptr;
            if (!pre) {
                CK(act_split_bf16x2_f32(A, M, K, K, tmp_planes, tmp_planes + (size_t)M * K, s));
                ah = tmp_planes; al = tmp_planes + (size_t)M * K;
            }
            return act_sgemm_nt_bf16x3_planes_f32(M, N, K, ah, al, W_hi, W_hi + wplane, emit ? nullptr : C, N, emit, emit ? emit + (size_t)M * N : nullptr, &e, s);
        }
        return gemm_nt(M, N, K, A, K, W, K, C, N, e, ws, wsb, s);
    };
    // the MLP pair goes through planes only when BOTH products take the split-bf16 kernel
    const bool mlp_planes = x3_ok && x3->fc1 && x3->fc2 && act_sgemm_nt_bf16x3_supported(TG, Hd, D) && act_sgemm_nt_bf16x3_supported(TG, D, Hd);
    act_gemm_epilogue_t e = epi0();
    e.bias = w.qkv_b ? w.qkv_b + D : nullptr;                                                   // K,V rows of the qkv Linear
    // (n1p_is_planes: the caller's prompt LayerNorm already left the rows as planes in the activation region)
    CK(linear(TP, 2 * D, D, n1p, w.qkv_w + (size_t)D * D, x3 ? x3->qkv + (size_t)D * D : nullptr, (size_t)3 * D * D, sv.kvp, e,
              (x3_ok && x3->n1p_is_planes) ? tmp_planes : nullptr));
    const bool ln1_planes = x3_ok && x3->qkv && act_sgemm_nt_bf16x3_supported(TG, 3 * D, D);          // LayerNorm-1 hands n1x on as planes (xin stays fp32: residual)
    if (ln1_planes) RUN(act_layernorm_fwd_planes_f32(x, pos, w.norm1_w, w.norm1_b, sv.xin, nullptr, tmp_planes, tmp_planes + (size_t)TG * D, keep ? sv.mean1 : nullptr,
                                                     keep ? sv.rstd1 : nullptr, TG, D, d.eps, s));
    else RUN(act_layernorm_fwd_f32(x, pos, w.norm1_w, w.norm1_b, sv.xin, sv.n1x, keep ? sv.mean1 : nullptr, keep ? sv.rstd1 : nullptr, TG, D, d.eps, s));
    e = epi0(); e.bias = w.qkv_b;
    CK(linear(TG, 3 * D, D, sv.n1x, w.qkv_w, x3 ? x3->qkv : nullptr, (size_t)3 * D * D, sv.qkvx, e, ln1_planes ? tmp_planes : nullptr));
    const bool att_planes = x3_ok && x3->proj && act_sgemm_nt_bf16x3_supported(TG, D, D);       // the attention output leaves its kernel as planes
    if (att_planes) RUN(act_attention_fwd_prefix_planes_f32(sv.kvp, P, sv.qkvx, G, keep ? sv.att : nullptr, tmp_planes, tmp_planes + (size_t)TG * D, keep ? sv.lse : nullptr,
                                                            B, H, hd, attn_scale(hd), s));
    else RUN(act_attention_fwd_prefix_f32(sv.kvp, P, sv.qkvx, G, sv.att, keep ? sv.lse : nullptr, B, H, hd, attn_scale(hd), s));
    e = epi0(); e.bias = w.proj_b; e.res = sv.xin; e.ldr = D;
    CK(linear(TG, D, D, sv.att, w.proj_w, x3 ? x3->proj : nullptr, (size_t)D * D, sv.x1, e, att_planes ? tmp_planes : nullptr));
    const bool ln2_planes = x3_ok && x3->fc1 && act_sgemm_nt_bf16x3_supported(TG, Hd, D);
    if (ln2_planes) RUN(act_layernorm_fwd_planes_f32(sv.x1, nullptr, w.norm2_w, w.norm2_b, nullptr, nullptr, tmp_planes, tmp_planes + (size_t)TG * D, keep ? sv.mean2 : nullptr,
                                                     keep ? sv.rstd2 : nullptr, TG, D, d.eps, s));
    else RUN(act_layernorm_fwd_f32(sv.x1, nullptr, w.norm2_w, w.norm2_b, nullptr, sv.n2, keep ? sv.mean2 : nullptr, keep ? sv.rstd2 : nullptr, TG, D, d.eps, s));
    e = epi0(); e.bias = w.fc1_b; e.act = ACT_EPI_GELU; e.aux = keep ? sv.hpre : nullptr; e.ldaux = Hd;
    CK(linear(TG, Hd, D, sv.n2, w.fc1_w, x3 ? x3->fc1 : nullptr, (size_t)Hd * D, sv.a, e, ln2_planes ? tmp_planes : nullptr, mlp_planes ? hid_planes : nullptr));
    e = epi0(); e.bias = w.fc2_b; e.res = sv.x1; e.ldr = D;
    // fc2 on the split-bf16 kernel ONLY with its operand already in hid_planes: a lone fc2 would split sv.a [TG, Hd] into tmp_planes, which holds
    // 2 * max(TG, TP) * D elements -- too small whenever Hd > D (reachable with a non-4x mlp_ratio, D % 128 == 0, Hd % 64 == 0, Hd % 128 != 0)
    CK(linear(TG, D, Hd, sv.a, w.fc2_w, (x3 && mlp_planes) ? x3->fc2 : nullptr, (size_t)D * Hd, out, e, mlp_planes ? hid_planes : nullptr));
    return 0;
}

}  // namespace

extern "C" {

int act_composite_shutdown(void) {
    std::lock_guard<std::mutex> g(g_ring_mu);
    int n = 0;
    for (auto& kv : g_rings)
        for (hipEvent_t e : kv.second.ev) { (void)hipEventDestroy(e); ++n; }
    g_rings.clear();
    return n;
}

int act_composite_collect_begin(void) {
    if (t_collect) return ACT_E_BADARG;
    t_collect = new std::vector<GemmShape>();
    return 0;
}
// -> number of recorded GEMMs (the first `max` are written to shapes [max][5] = a_kmajor, b_kmajor, M, N, K)
int act_composite_collect_end(int* shapes, int max) {
    if (!t_collect) return ACT_E_BADARG;
    const int n = (int)t_collect->size();
    for (int i = 0; i < n && i < max && shapes; ++i) {
        const GemmShape& g = (*t_collect)[i];
        shapes[5 * i] = g.ak; shapes[5 * i + 1] = g.bk; shapes[5 * i + 2] = g.M; shapes[5 * i + 3] = g.N; shapes[5 * i + 4] = g.K;
    }
    delete t_collect; t_collect = nullptr;
    return n;
}

// ============================================================================================== Transformer block
size_t act_block_saved_floats(const act_block_dims_t* d) {
    if (bad_dims(d)) return 0;
    BlockSaved sv; return carve_block(nullptr, *d, sv);
}
size_t act_block_bwd_scratch_floats(const act_block_dims_t* d) {
    if (bad_dims(d)) return 0;
    BlockBwdScratch sc; return carve_block_bwd(nullptr, *d, sc);
}

int act_block_fwd_f32(const act_block_dims_t* d, const act_block_params_t* w, const float* x, const float* pos, const float* gate1,
                      const float* gate2, int keep_for_backward, float* saved, float* out, float* ws, size_t wsb, act_stream_t stream) {
    if (!w || !x || !saved || !out) return ACT_E_NULLPTR;
    if (bad_dims(d)) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int B = d->B, S = d->S, D = d->D, H = d->heads, hd = D / H, Hd = d->hidden, T = B * S;
    const bool keep = keep_for_backward != 0;
    BlockSaved sv; carve_block(saved, *d, sv);
    RUN(act_layernorm_fwd_f32(x, pos, w->norm1_w, w->norm1_b, sv.xin, sv.n1, keep ? sv.mean1 : nullptr, keep ? sv.rstd1 : nullptr, T, D, d->eps, s));
    act_gemm_epilogue_t e = epi0(); e.bias = w->qkv_b;
    CK(gemm_nt(T, 3 * D, D, sv.n1, D, w->qkv_w, D, sv.qkv, 3 * D, e, ws, wsb, s));
    RUN(act_attention_fwd_f32(sv.qkv, sv.att, keep ? sv.lse : nullptr, B, S, H, hd, attn_scale(hd), s));
    e = epi0(); e.bias = w->proj_b; e.rowscale = gate1; e.rows_per_scale = S; e.res = sv.xin; e.ldr = D;
    CK(gemm_nt(T, D, D, sv.att, D, w->proj_w, D, sv.x1, D, e, ws, wsb, s));
    RUN(act_layernorm_fwd_f32(sv.x1, nullptr, w->norm2_w, w->norm2_b, nullptr, sv.n2, keep ? sv.mean2 : nullptr, keep ? sv.rstd2 : nullptr, T, D, d->eps, s));
    e = epi0(); e.bias = w->fc1_b; e.act = ACT_EPI_GELU; e.aux = keep ? sv.hpre : nullptr; e.ldaux = Hd;
    CK(gemm_nt(T, Hd, D, sv.n2, D, w->fc1_w, D, sv.a, Hd, e, ws, wsb, s));
    e = epi0(); e.bias = w->fc2_b; e.rowscale = gate2; e.rows_per_scale = S; e.res = sv.x1; e.ldr = D;
    CK(gemm_nt(T, D, Hd, sv.a, Hd, w->fc2_w, Hd, out, D, e, ws, wsb, s));
    return 0;
}

int act_block_bwd_f32(const act_block_dims_t* d, const act_block_params_t* w, const float* gate1, const float* gate2, const float* saved,
                      const float* dout, float* dx, const act_block_grads_t* g, float* scratch, float* ws, size_t wsb, float* sws,
                      size_t swsb, act_stream_t stream, act_stream_t side_stream) {
    if (!w || !saved || !dout || !dx || !scratch) return ACT_E_NULLPTR;
    if (bad_dims(d)) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const bool fork = g && side_stream && side_stream != stream && sws;
    hipStream_t ss = fork ? (hipStream_t)side_stream : s;           // stream of the weight-gradient work
    float* wws = fork ? sws : ws; const size_t wwsb = fork ? swsb : wsb;
    const int B = d->B, S = d->S, D = d->D, H = d->heads, hd = D / H, Hd = d->hidden, T = B * S;
    BlockSaved sv; carve_block(const_cast<float*>(saved), *d, sv);
    BlockBwdScratch sc; carve_block_bwd(scratch, *d, sc);

    // dW = dy^T . x and db = column sums of dy, after everything enqueued so far on `stream`
    auto wgrad = [&](const float* dy, int N, const float* xop, int K, float* dw, float* db) -> int {
        if (!g) return 0;
        if (fork) CK(order_after(ss, s));
        if (dw) CK(gemm_tn(N, K, T, dy, N, xop, K, dw, K, wws, wwsb, ss));
        if (db) CK(colsum(dy, T, N, db, wws, wwsb, ss));
        return 0;
    };
    // ... or two Linears at a time: ONE grouped launch (gemm_grouped.hip) computes both weight gradients and both bias gradients
    // (+ one reduction launch for the K ranges) instead of 2 GEMMs + 2 split-K reductions + 4 column-sum launches.
    static const bool grouped_env = [] { const char* e = getenv("ACT_GROUPED_DW"); return !(e && e[0] == '0'); }();
    const bool grouped = grouped_env && g && g->fc2_w && g->fc1_w && g->proj_w && g->qkv_w && (D % 128) == 0 && (Hd % 128) == 0 && (T % 16) == 0 && wws;
    auto wgrad2 = [&](const act_gemm_tn_problem_t* pr) -> int {
        if (t_collect) return 0;
        if (fork) CK(order_after(ss, s));
        // the grouped launch wants 16-byte aligned operands and leading dimensions % 4 == 0 (a gradient or activation pointer at an odd offset inside a
        // larger allocation): anything else takes the two single products, whose ragged-edge kernels handle any alignment (round-3 advisor finding)
        bool ok = true;
        for (int i = 0; i < 2; ++i)
            ok = ok && (((reinterpret_cast<uintptr_t>(pr[i].A) | reinterpret_cast<uintptr_t>(pr[i].B) | reinterpret_cast<uintptr_t>(pr[i].C)) & 15) == 0) &&
                 ((pr[i].lda | pr[i].ldb | pr[i].ldc) & 3) == 0;
        if (!ok) {
            for (int i = 0; i < 2; ++i) {
                if (pr[i].C) CK(gemm_tn(pr[i].M, pr[i].N, T, pr[i].A, pr[i].lda, pr[i].B, pr[i].ldb, pr[i].C, pr[i].ldc, wws, wwsb, ss));
                if (pr[i].bias_out) CK(colsum(pr[i].A, T, pr[i].M, pr[i].bias_out, wws, wwsb, ss));
            }
            return 0;
        }
        // K ranges: the count act_sgemm_tn_grouped_splits asks for, capped by the workspace the caller provided -- the host path (kernels.gemm_tn_grouped)
        // applies the same cap (kernels._WS_BYTES) instead of growing its workspace, so both paths use the same summation order
        int sp = act_sgemm_tn_grouped_splits(pr, 2, T);
        while (sp > 1 && act_sgemm_tn_grouped_workspace(pr, 2, T, sp) > wwsb) --sp;
        return act_sgemm_tn_grouped_f32(pr, 2, T, sp, wws, wwsb, ss);
    };

    const float* dy2 = dout;
    if (gate2) { RUN(act_scale_rows_f32(dout, gate2, T, D, S, sc.dy2, s)); dy2 = sc.dy2; }
    if (!grouped) CK(wgrad(dy2, D, sv.a, Hd, g ? g->fc2_w : nullptr, g ? g->fc2_b : nullptr));
    act_gemm_epilogue_t e = epi0(); e.act = ACT_EPI_MUL_GELU_GRAD; e.aux = sv.hpre; e.ldaux = Hd;
    CK(gemm_nn(T, Hd, D, dy2, D, w->fc2_w, Hd, sc.dh, Hd, e, ws, wsb, s));
    if (grouped) {
        const act_gemm_tn_problem_t pr[2] = {{dy2, D, sv.a, Hd, g->fc2_w, Hd, D, Hd, g->fc2_b}, {sc.dh, Hd, sv.n2, D, g->fc1_w, D, Hd, D, g->fc1_b}};
        CK(wgrad2(pr));
    } else {
        CK(wgrad(sc.dh, Hd, sv.n2, D, g ? g->fc1_w : nullptr, g ? g->fc1_b : nullptr));
    }
    CK(gemm_nn(T, D, Hd, sc.dh, Hd, w->fc1_w, D, sc.dn2, D, epi0(), ws, wsb, s));
    RUN(act_layernorm_bwd_f32(sc.dn2, sv.x1, w->norm2_w, sv.mean2, sv.rstd2, dout, sc.dx1, g ? g->norm2_w : nullptr, g ? g->norm2_b : nullptr, 0,
                             ws, wsb, T, D, s));
    const float* dy1 = sc.dx1;
    if (gate1) { RUN(act_scale_rows_f32(sc.dx1, gate1, T, D, S, sc.dy1, s)); dy1 = sc.dy1; }
    if (!grouped) CK(wgrad(dy1, D, sv.att, D, g ? g->proj_w : nullptr, g ? g->proj_b : nullptr));
    CK(gemm_nn(T, D, D, dy1, D, w->proj_w, D, sc.datt, D, epi0(), ws, wsb, s));
    RUN(act_attention_bwd_f32(sv.qkv, sv.att, sc.datt, sv.lse, sc.dqkv, B, S, H, hd, attn_scale(hd), s));
    if (grouped) {
        const act_gemm_tn_problem_t pr[2] = {{dy1, D, sv.att, D, g->proj_w, D, D, D, g->proj_b},
                                             {sc.dqkv, 3 * D, sv.n1, D, g->qkv_w, D, 3 * D, D, w->qkv_b ? g->qkv_b : nullptr}};
        CK(wgrad2(pr));
    } else {
        CK(wgrad(sc.dqkv, 3 * D, sv.n1, D, g ? g->qkv_w : nullptr, (g && w->qkv_b) ? g->qkv_b : nullptr));
    }
    CK(gemm_nn(T, D, 3 * D, sc.dqkv, 3 * D, w->qkv_w, D, sc.dn1, D, epi0(), ws, wsb, s));
    RUN(act_layernorm_bwd_f32(sc.dn1, sv.xin, w->norm1_w, sv.mean1, sv.rstd1, sc.dx1, dx, g ? g->norm1_w : nullptr, g ? g->norm1_b : nullptr, 0,
                             ws, wsb, T, D, s));
    if (fork) CK(order_after(s, ss));
    return 0;
}

// ============================================================================================== a stack of blocks (TransformerEncoder / Decoder loop)
static bool bad_stack(const act_block_dims_t* d, const act_block_stack_t* st) { return bad_dims(d) || !st || st->depth <= 0 || !st->blocks; }
size_t act_block_stack_saved_floats(const act_block_dims_t* d, int depth, int keep_for_backward) {
    if (bad_dims(d) || depth <= 0) return 0;
    // per block the slab of act_block_fwd_f32 (ONE slab re-used by every block when nothing is kept) + two [T, D] buffers for the hidden state between blocks
    return act_block_saved_floats(d) * (size_t)(keep_for_backward ? depth : 1) + 2 * (size_t)d->B * d->S * d->D;
}
size_t act_block_stack_bwd_scratch_floats(const act_block_dims_t* d, int depth) {
    if (bad_dims(d) || depth <= 0) return 0;
    return act_block_bwd_scratch_floats(d) + 2 * (size_t)d->B * d->S * d->D;
}
int act_block_stack_fwd_f32(const act_block_dims_t* d, const act_block_stack_t* st, const float* x, const float* pos, int keep_for_backward,
                            float* saved, float* out, float* ws, size_t wsb, act_stream_t stream) {
    if (!x || !saved || !out) return ACT_E_NULLPTR;
    if (bad_stack(d, st)) return ACT_E_BADARG;
    const size_t per = act_block_saved_floats(d), TD = (size_t)d->B * d->S * d->D;
    const int L = st->depth;
    float* hid[2] = {saved + per * (size_t)(keep_for_backward ? L : 1), saved + per * (size_t)(keep_for_backward ? L : 1) + TD};
    const float* cur = x;
    for (int l = 0; l < L; ++l) {
        float* o = (l == L - 1) ? out : hid[l & 1];
        CK(act_block_fwd_f32(d, &st->blocks[l], cur, pos, st->gate1 ? st->gate1[l] : nullptr, st->gate2 ? st->gate2[l] : nullptr, keep_for_backward,
                             saved + (keep_for_backward ? per * (size_t)l : 0), o, ws, wsb, stream));
        cur = o;
    }
    return 0;
}
int act_block_stack_bwd_f32(const act_block_dims_t* d, const act_block_stack_t* st, const float* saved, const float* dout, float* dx, float* dpos,
                            const float* dpos_in, const act_block_grads_t* grads, float* scratch, float* ws, size_t wsb, float* sws, size_t swsb,
                            act_stream_t stream, act_stream_t side_stream) {
    if (!saved || !dout || !dx || !scratch) return ACT_E_NULLPTR;
    if (bad_stack(d, st)) return ACT_E_BADARG;
    if (dpos_in && !dpos) return ACT_E_NULLPTR;
    const size_t per = act_block_saved_floats(d), TD = (size_t)d->B * d->S * d->D;
    const int L = st->depth;
    float* blk_scratch = scratch;
    float* hid[2] = {scratch + act_block_bwd_scratch_floats(d), scratch + act_block_bwd_scratch_floats(d) + TD};
    const float* cur = dout;
    // gradient of the shared pos: ((dx_{L-1} + dx_{L-2}) + dx_{L-3}) + ... in the order the blocks finish (what an autograd engine's input buffer does).
    // dpos_in (round 6) = the sum the DEEPER chunks of the same stack already folded (blocks L, L+1, ... of the whole stack): the chain continues through
    // it, ((dpos_in + dx_{L-1}) + dx_{L-2}) + ..., so a stack differentiated in chunks associates exactly like the unchunked one.
    const float* acc = dpos_in;
    for (int l = L - 1; l >= 0; --l) {
        float* o = (l == 0) ? dx : hid[l & 1];
        CK(act_block_bwd_f32(d, &st->blocks[l], st->gate1 ? st->gate1[l] : nullptr, st->gate2 ? st->gate2[l] : nullptr, saved + per * (size_t)l, cur, o,
                             grads ? &grads[l] : nullptr, blk_scratch, ws, wsb, sws, swsb, stream, side_stream));
        if (dpos) {
            if (acc) { RUN(act_add_f32(acc, o, dpos, (long long)TD, stream)); acc = dpos; }
            else acc = o;                                        // first term: nothing to add yet (depth 1 without dpos_in: the caller uses dx as dpos)
        }
        cur = o;
    }
    return 0;
}

// ============================================================================================== prefix block (prompts = keys / values only)
size_t act_prefix_block_saved_floats(const act_block_dims_t* d, int P) {
    if (bad_dims(d) || P < 0) return 0;
    PrefixSaved sv; return carve_prefix(nullptr, *d, P, sv);
}
size_t act_prefix_block_bwd_scratch_floats(const act_block_dims_t* d, int P) {
    if (bad_dims(d) || P < 0) return 0;
    PrefixBwdScratch sc; return carve_prefix_bwd(nullptr, *d, P, sc);
}

int act_prefix_block_fwd_f32(const act_block_dims_t* d, int P, const act_block_params_t* w, const float* x, const float* pos, const float* prm,
                             const float* n1p_in, int keep_for_backward, float* saved, float* out, float* ws, size_t wsb, act_stream_t stream) {
    if (!w || !x || !saved || !out || (!prm && !n1p_in)) return ACT_E_NULLPTR;
    if (bad_dims(d) || P <= 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const bool keep = keep_for_backward != 0;
    PrefixSaved sv; carve_prefix(saved, *d, P, sv);
    const float* n1p = n1p_in;
    if (!n1p) {
        RUN(act_layernorm_fwd_f32(prm, nullptr, w->norm1_w, w->norm1_b, nullptr, sv.n1p, keep ? sv.meanp : nullptr, keep ? sv.rstdp : nullptr,
                                 d->B * P, d->D, d->eps, s));
        n1p = sv.n1p;
    }
    return prefix_block_core(*d, P, *w, x, pos, n1p, keep, sv, out, ws, wsb, s);
}

// OPT-IN: the same forward with the block's four (frozen) weights given as (hi, lo) bf16 planes: every product whose shape the split-bf16 kernel takes runs there
// (w_planes[0..3] = hi planes of qkv_w, proj_w, fc1_w, fc2_w, lo plane behind each; a_planes as in act_vit_bf16x3_t).  The backward is unchanged (f32).
int act_prefix_block_fwd_bf16x3_f32(const act_block_dims_t* d, int P, const act_block_params_t* w, const act_vit_bf16x3_t* x3, const float* x, const float* pos,
                                    const float* prm, int keep_for_backward, float* saved, float* out, float* ws, size_t wsb, act_stream_t stream) {
    if (!w || !x || !saved || !out || !prm || !x3 || !x3->w_planes || !x3->a_planes) return ACT_E_NULLPTR;
    if (bad_dims(d) || P <= 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const bool keep = keep_for_backward != 0;
    PrefixSaved sv; carve_prefix(saved, *d, P, sv);
    X3Block xb{x3->w_planes[0], x3->w_planes[1], x3->w_planes[2], x3->w_planes[3], x3->a_planes, x3->a_planes_elems, false};
    const int TP = d->B * P, D = d->D;
    if (xb.qkv && x3_fits(*d, P, xb.a_elems) && act_sgemm_nt_bf16x3_supported(TP, 2 * D, D)) {
        uint16_t* tp = xb.a_planes + (size_t)2 * d->B * d->S * d->hidden;
        RUN(act_layernorm_fwd_planes_f32(prm, nullptr, w->norm1_w, w->norm1_b, nullptr, nullptr, tp, tp + (size_t)TP * D, keep ? sv.meanp : nullptr,
                                        keep ? sv.rstdp : nullptr, TP, D, d->eps, s));
        xb.n1p_is_planes = true;
    } else {
        RUN(act_layernorm_fwd_f32(prm, nullptr, w->norm1_w, w->norm1_b, nullptr, sv.n1p, keep ? sv.meanp : nullptr, keep ? sv.rstdp : nullptr, TP, D, d->eps, s));
    }
    return prefix_block_core(*d, P, *w, x, pos, sv.n1p, keep, sv, out, ws, wsb, s, &xb);
}

int act_prefix_block_bwd_f32(const act_block_dims_t* d, int P, const act_block_params_t* w, const float* prm, const float* saved,
                             const float* dout, float* dx, float* dprm, float* scratch, float* ws, size_t wsb, act_stream_t stream) {
    if (!w || !prm || !saved || !dout || !dx || !dprm || !scratch) return ACT_E_NULLPTR;
    if (bad_dims(d) || P <= 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int B = d->B, G = d->S, D = d->D, H = d->heads, hd = D / H, Hd = d->hidden, TG = B * G, TP = B * P;
    PrefixSaved sv; carve_prefix(const_cast<float*>(saved), *d, P, sv);
    PrefixBwdScratch sc; carve_prefix_bwd(scratch, *d, P, sc);
    act_gemm_epilogue_t e = epi0(); e.act = ACT_EPI_MUL_GELU_GRAD; e.aux = sv.hpre; e.ldaux = Hd;
    CK(gemm_nn(TG, Hd, D, dout, D, w->fc2_w, Hd, sc.dh, Hd, e, ws, wsb, s));
    CK(gemm_nn(TG, D, Hd, sc.dh, Hd, w->fc1_w, D, sc.dn2, D, epi0(), ws, wsb, s));
    RUN(act_layernorm_bwd_f32(sc.dn2, sv.x1, w->norm2_w, sv.mean2, sv.rstd2, dout, sc.dx1, nullptr, nullptr, 0, nullptr, 0, TG, D, s));
    CK(gemm_nn(TG, D, D, sc.dx1, D, w->proj_w, D, sc.datt, D, epi0(), ws, wsb, s));
    RUN(act_attention_bwd_prefix_f32(sv.kvp, P, sv.qkvx, G, sv.att, sc.datt, sv.lse, sc.dkvp, sc.dqkvx, B, H, hd, attn_scale(hd), s));
    CK(gemm_nn(TG, D, 3 * D, sc.dqkvx, 3 * D, w->qkv_w, D, sc.dn1x, D, epi0(), ws, wsb, s));
    RUN(act_layernorm_bwd_f32(sc.dn1x, sv.xin, w->norm1_w, sv.mean1, sv.rstd1, sc.dx1, dx, nullptr, nullptr, 0, nullptr, 0, TG, D, s));
    CK(gemm_nn(TP, D, 2 * D, sc.dkvp, 2 * D, w->qkv_w + (size_t)D * D, D, sc.dn1p, D, epi0(), ws, wsb, s));
    RUN(act_layernorm_bwd_f32(sc.dn1p, prm, w->norm1_w, sv.meanp, sv.rstdp, nullptr, dprm, nullptr, nullptr, 0, nullptr, 0, TP, D, s));
    return 0;
}

// OPT-IN: the same backward with the five input-gradient products on the split-bf16 kernel.  wT_planes[0..4] = hi planes (lo plane `numel` behind each) of the
// TRANSPOSED frozen weights, which is what turns dY . W into the kernel's A . B^T form: fc2_w^T [Hd][D], fc1_w^T [D][Hd], proj_w^T [D][D], qkv_w^T [D][3D] and
// (qkv_w rows D..3D)^T [D][2D].  Each incoming gradient is split into planes first (dh leaves the fc2 product's epilogue as planes); LayerNorm / attention
// backward are the f32 kernels.  A product whose shape the kernel does not take stays on the f32 kernels.
int act_prefix_block_bwd_bf16x3_f32(const act_block_dims_t* d, int P, const act_block_params_t* w, const act_vit_bf16x3_t* x3, const float* prm, const float* saved,
                                    const float* dout, float* dx, float* dprm, float* scratch, float* ws, size_t wsb, act_stream_t stream) {
    if (!w || !prm || !saved || !dout || !dx || !dprm || !scratch || !x3 || !x3->w_planes || !x3->a_planes) return ACT_E_NULLPTR;
    if (bad_dims(d) || P <= 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int B = d->B, G = d->S, D = d->D, H = d->heads, hd = D / H, Hd = d->hidden, TG = B * G, TP = B * P;
    if (!x3_fits(*d, P, x3->a_planes_elems) || (size_t)TP * 2 * D > (size_t)TG * Hd || 3 * D > Hd)               // the staging regions below would not hold it
        return act_prefix_block_bwd_f32(d, P, w, prm, saved, dout, dx, dprm, scratch, ws, wsb, stream);
    PrefixSaved sv; carve_prefix(const_cast<float*>(saved), *d, P, sv);
    PrefixBwdScratch sc; carve_prefix_bwd(scratch, *d, P, sc);
    const uint16_t *fc2T = x3->w_planes[0], *fc1T = x3->w_planes[1], *projT = x3->w_planes[2], *qkvT = x3->w_planes[3], *kvT = x3->w_planes[4];
    uint16_t* big = x3->a_planes;                                   // 2 TG Hd: planes of dh, later of dqkvx [TG, 3D], later of dkvp [TP, 2D]
    uint16_t* small = x3->a_planes + (size_t)2 * TG * Hd;           // 2 max(TG, TP) D: planes of dout, later of dx1
    // dIn [M, N] = dOut [M, K] . W [K, N]  (W^T as planes [N][K]); `pre`: dOut is already planes at `stage`; `emit`: the result leaves as planes only
    auto dgrad = [&](int M, int N, int K, const float* dOut, const float* W, const uint16_t* WT, float* dIn, const act_gemm_epilogue_t& e, uint16_t* stage,
                     bool pre = false, uint16_t* emit = nullptr) -> int {
        if (WT && act_sgemm_nt_bf16x3_supported(M, N, K)) {
            if (t_collect) return 0;
            if (!pre) CK(act_split_bf16x2_f32(dOut, M, K, K, stage, stage + (size_t)M * K, s));
            return act_sgemm_nt_bf16x3_planes_f32(M, N, K, stage, stage + (size_t)M * K, WT, WT + (size_t)N * K, emit ? nullptr : dIn, N, emit,
                                                  emit ? emit + (size_t)M * N : nullptr, &e, s);
        }
        return gemm_nn(M, N, K, dOut, K, W, N, dIn, N, e, ws, wsb, s);
    };
    const bool mlp_planes = fc2T && fc1T && act_sgemm_nt_bf16x3_supported(TG, Hd, D) && act_sgemm_nt_bf16x3_supported(TG, D, Hd);
    act_gemm_epilogue_t e = epi0(); e.act = ACT_EPI_MUL_GELU_GRAD; e.aux = sv.hpre; e.ldaux = Hd;
    CK(dgrad(TG, Hd, D, dout, w->fc2_w, fc2T, sc.dh, e, small, false, mlp_planes ? big : nullptr));
    CK(dgrad(TG, D, Hd, sc.dh, w->fc1_w, fc1T, sc.dn2, epi0(), big, mlp_planes));
    RUN(act_layernorm_bwd_f32(sc.dn2, sv.x1, w->norm2_w, sv.mean2, sv.rstd2, dout, sc.dx1, nullptr, nullptr, 0, nullptr, 0, TG, D, s));
    CK(dgrad(TG, D, D, sc.dx1, w->proj_w, projT, sc.datt, epi0(), small));
    RUN(act_attention_bwd_prefix_f32(sv.kvp, P, sv.qkvx, G, sv.att, sc.datt, sv.lse, sc.dkvp, sc.dqkvx, B, H, hd, attn_scale(hd), s));
    CK(dgrad(TG, D, 3 * D, sc.dqkvx, w->qkv_w, qkvT, sc.dn1x, epi0(), big));
    RUN(act_layernorm_bwd_f32(sc.dn1x, sv.xin, w->norm1_w, sv.mean1, sv.rstd1, sc.dx1, dx, nullptr, nullptr, 0, nullptr, 0, TG, D, s));
    CK(dgrad(TP, D, 2 * D, sc.dkvp, w->qkv_w + (size_t)D * D, kvT, sc.dn1p, epi0(), big));
    RUN(act_layernorm_bwd_f32(sc.dn1p, prm, w->norm1_w, sv.meanp, sv.rstdp, nullptr, dprm, nullptr, nullptr, 0, nullptr, 0, TP, D, s));
    return 0;
}

// ============================================================================================== frozen prompt-tuned Transformer (teacher)
static size_t carve_vit(float* base, const act_prefix_vit_t& m, float*& pos_h, float*& pos, float*& xa, float*& xb, float*& n1p, float*& blk,
                        float*& feat) {
    const size_t TG = (size_t)m.B * m.G, TP = (size_t)m.B * m.P, D = m.D;
    act_block_dims_t d{m.B, m.G, m.D, m.heads, m.hidden, m.eps};
    PrefixSaved sv;
    Carver c(base);
    pos_h = c.take(TG * m.pos_hidden); pos = c.take(TG * D); xa = c.take(TG * D); xb = c.take(TG * D); n1p = c.take(TP * D); feat = c.take(TG * D);
    blk = c.take(carve_prefix(nullptr, d, m.P, sv));
    return c.used;
}
static bool bad_vit(const act_prefix_vit_t* m) {
    return !m || m->B <= 0 || m->P <= 0 || m->G <= 0 || m->D <= 0 || m->heads <= 0 || m->D % m->heads || m->hidden <= 0 || m->depth <= 0 ||
           m->tokens_dims <= 0 || m->pos_hidden <= 0;
}
size_t act_prefix_vit_scratch_floats(const act_prefix_vit_t* m) {
    if (bad_vit(m)) return 0;
    float *a, *b, *c, *d, *e, *f, *g;
    return carve_vit(nullptr, *m, a, b, c, d, e, f, g);
}

static int prefix_vit_fwd(const act_prefix_vit_t* m, const act_vit_bf16x3_t* x3, const float* tokens, const float* center, float* out, float* scratch,
                          float* ws, size_t wsb, act_stream_t stream);
int act_prefix_vit_fwd_f32(const act_prefix_vit_t* m, const float* tokens, const float* center, float* out, float* scratch, float* ws,
                           size_t wsb, act_stream_t stream) {
    return prefix_vit_fwd(m, nullptr, tokens, center, out, scratch, ws, wsb, stream);
}
int act_prefix_vit_fwd_bf16x3_f32(const act_prefix_vit_t* m, const act_vit_bf16x3_t* x3, const float* tokens, const float* center, float* out,
                                  float* scratch, float* ws, size_t wsb, act_stream_t stream) {
    if (!x3 || !x3->w_planes || !x3->a_planes) return ACT_E_NULLPTR;
    return prefix_vit_fwd(m, x3, tokens, center, out, scratch, ws, wsb, stream);
}
static int prefix_vit_fwd(const act_prefix_vit_t* m, const act_vit_bf16x3_t* x3, const float* tokens, const float* center, float* out, float* scratch,
                          float* ws, size_t wsb, act_stream_t stream) {
    if (bad_vit(m)) return ACT_E_BADARG;
    if (!tokens || !center || !out || !scratch || !m->blocks || !m->prompt_tok || !m->prompt_pos) return ACT_E_NULLPTR;
    hipStream_t s = (hipStream_t)stream;
    const int TG = m->B * m->G, D = m->D;
    float *pos_h, *pos, *xa, *xb, *n1p, *blk, *feat;
    carve_vit(scratch, *m, pos_h, pos, xa, xb, n1p, blk, feat);
    const act_block_dims_t d{m->B, m->G, m->D, m->heads, m->hidden, m->eps};
    PrefixSaved sv; carve_prefix(blk, d, m->P, sv);
    // pos = visual_pos_embed(center): Linear(3, pos_hidden) - GELU - Linear(pos_hidden, D)   (models/dvae.py:413-417)
    act_gemm_epilogue_t e = epi0(); e.bias = m->pos_b0; e.act = ACT_EPI_GELU;
    CK(gemm_nt(TG, m->pos_hidden, 3, center, 3, m->pos_w0, 3, pos_h, m->pos_hidden, e, ws, wsb, s));
    e = epi0(); e.bias = m->pos_b1;
    CK(gemm_nt(TG, D, m->pos_hidden, pos_h, m->pos_hidden, m->pos_w1, m->pos_hidden, pos, D, e, ws, wsb, s));
    e = epi0(); e.bias = m->pre_b;
    CK(gemm_nt(TG, D, m->tokens_dims, tokens, m->tokens_dims, m->pre_w, m->tokens_dims, xa, D, e, ws, wsb, s));
    float* cur = xa; float* nxt = xb;
    for (int i = 0; i < m->depth; ++i) {
        const act_block_params_t& w = m->blocks[i];
        const uint64_t seed = (m->seed_base + 7919ull * (uint64_t)(i + 1)) & ((1ull << 62) - 1);
        X3Block xb{};
        if (x3) xb = X3Block{x3->w_planes[4 * i], x3->w_planes[4 * i + 1], x3->w_planes[4 * i + 2], x3->w_planes[4 * i + 3], x3->a_planes, x3->a_planes_elems, false};
        if (x3 && xb.qkv && x3_fits(d, m->P, x3->a_planes_elems) && act_sgemm_nt_bf16x3_supported(m->B * m->P, 2 * D, D)) {
            uint16_t* tp = x3->a_planes + (size_t)2 * TG * m->hidden;                         // the activation region (x3_fits)
            RUN(act_prompt_layernorm_fwd_planes_f32(m->prompt_tok[i], m->prompt_pos[i], m->B, m->P, D, m->drop_p, seed, m->seed_dev, w.norm1_w, w.norm1_b,
                                                   m->eps, tp, tp + (size_t)m->B * m->P * D, s));
            xb.n1p_is_planes = true;
        } else
        RUN(act_prompt_layernorm_fwd_f32(m->prompt_tok[i], m->prompt_pos[i], m->B, m->P, D, m->drop_p, seed, m->seed_dev, w.norm1_w, w.norm1_b,
                                        m->eps, n1p, s));
        CK(prefix_block_core(d, m->P, w, cur, pos, n1p, false, sv, nxt, ws, wsb, s, x3 ? &xb : nullptr));
        float* t = cur; cur = nxt; nxt = t;
    }
    RUN(act_layernorm_fwd_f32(cur, nullptr, m->norm_w, m->norm_b, nullptr, feat, nullptr, nullptr, TG, D, m->eps, s));
    e = epi0(); e.bias = m->post_b;
    CK(gemm_nt(TG, m->tokens_dims, D, feat, D, m->post_w, D, out, m->tokens_dims, e, ws, wsb, s));
    return 0;
}

// ============================================================================================== mini-PointNet (Encoder)
// Two launch schedules, chosen from the dimensions alone (so saved / scratch sizes are a function of the dims):
//  * fused (n in {32, 64}, rows % 128 == 0, C % 64 == 0; ACT_PN_FUSE=0 disables): BatchNorm + ReLU are applied while the NEXT conv stages
//    its A operand (a1 / a3 never exist), BatchNorm-2 statistics come out of the epilogue of the conv that produces its input, both
//    max-pools are epilogues (the 512 -> C conv does not even store its output), and the weight gradients of the convs behind a
//    BatchNorm recompute the activated operand on load.  Per forward: 2 statistics passes, 2 apply passes, 2 max-pool passes and the
//    h4 store (1.4 ms of the 35 ms Stage-II step for the two encoders) disappear.
//  * plain: one kernel per layer (tiny test geometries, odd channel counts).
namespace {
struct PnSaved { float *h1, *a1, *h2, *fg, *gw, *h3, *a3, *h4, *st1, *st2, *tstats; int32_t *arg1, *arg2; };   // st* = mean | rstd | scale | shift
bool pn_fused(const act_pointnet_dims_t& d) {
    static const bool on = [] { const char* e = getenv("ACT_PN_FUSE"); return !(e && e[0] == '0'); }();
    const long long R = (long long)d.BG * d.n;
    return on && (d.n == 32 || d.n == 64) && R % 128 == 0 && d.C % 64 == 0;
}
// fused schedule with C % 128 == 0: the scattered gradient of the second max-pool (dh4, [R][C]) is never written -- the two GEMMs that
// consume it generate it from (dout, arg2) while they stage their A operand, and the first pool's scatter-add is an epilogue of the GEMM
// that produces dh2
bool pn_pool_bwd_on_load(const act_pointnet_dims_t& d) {
    static const bool on = [] { const char* e = getenv("ACT_PN_POOL_BWD_FUSE"); return !(e && e[0] == '0'); }();
    return on && pn_fused(d) && d.C % 128 == 0;
}
// ... and walked sparsely instead of fed to a dense GEMM (pool_bwd.hip); ACT_PN_POOL_BWD_SPARSE=0 keeps the dense on-load products
bool pn_pool_bwd_sparse() {
    static const bool on = [] { const char* e = getenv("ACT_PN_POOL_BWD_SPARSE"); return !(e && e[0] == '0'); }();
    return on;
}
bool pn_pool_bwd_live() {                                                        // ACT_POOL_BWD_LIVE=0: every group is walked / written / read
    static const bool on = [] { const char* e = getenv("ACT_POOL_BWD_LIVE"); return !(e && e[0] == '0'); }();
    return on;
}
size_t carve_pn(float* base, const act_pointnet_dims_t& d, PnSaved& sv) {
    const size_t R = (size_t)d.BG * d.n, BG = d.BG, C = d.C;
    const bool fused = pn_fused(d);
    Carver c(base);
    sv.h1 = c.take(R * 128); sv.h2 = c.take(R * 256); sv.fg = c.take(BG * 256); sv.gw = c.take(BG * 512); sv.h3 = c.take(R * 512);
    sv.st1 = c.take(4 * 128); sv.st2 = c.take(4 * 512);
    sv.arg1 = reinterpret_cast<int32_t*>(c.take(BG * 256)); sv.arg2 = reinterpret_cast<int32_t*>(c.take(BG * C));
    if (fused) { sv.a1 = sv.a3 = sv.h4 = nullptr; sv.tstats = c.take(act_sgemm_fx_tile_stats_floats((int)R, 512)); }
    else { sv.a1 = c.take(R * 128); sv.a3 = c.take(R * 512); sv.h4 = c.take(R * C); sv.tstats = nullptr; }
    return c.used;
}
struct PnBwdScratch { float *dh4, *da3, *dh3, *dgw, *dh2, *dfg, *da1, *dh1, *act; int32_t* live; };
size_t carve_pn_bwd(float* base, const act_pointnet_dims_t& d, PnBwdScratch& sc) {
    const size_t R = (size_t)d.BG * d.n, BG = d.BG, C = d.C;
    Carver c(base);
    sc.dh4 = pn_pool_bwd_on_load(d) ? nullptr : c.take(R * C); sc.da3 = c.take(R * 512); sc.dh3 = c.take(R * 512); sc.dgw = c.take(BG * 512); sc.dh2 = c.take(R * 256);
    sc.dfg = c.take(BG * 256); sc.da1 = c.take(R * 128); sc.dh1 = c.take(R * 128);
    sc.act = (pn_fused(d) && d.C % 128 != 0) ? c.take(R * 512) : nullptr;      // a3 rebuilt for the one weight gradient the fused TN kernel cannot take
    sc.live = reinterpret_cast<int32_t*>(c.take(BG));                            // groups with a non-zero gradient row (Stage II: the visible patches)
    return c.used;
}
bool bad_pn(const act_pointnet_dims_t* d) { return !d || d->BG <= 0 || d->n <= 0 || d->C <= 0 || (d->C & 3); }

int gemm_fx(int ak, int bk, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, const act_gemm_epilogue_t& e,
            const act_gemm_fx_t& fx, float* ws, size_t wsb, hipStream_t s) {
    if (collecting(ak, bk, M, N, K)) return 0;          // (fused kernels pick their own tile; recorded for completeness)
    return act_sgemm_fx_f32(ak, bk, M, N, K, A, lda, B, ldb, C, ldc, &e, &fx, ws, wsb, s);
}
}  // namespace

size_t act_pointnet_saved_floats(const act_pointnet_dims_t* d) { if (bad_pn(d)) return 0; PnSaved sv; return carve_pn(nullptr, *d, sv); }
size_t act_pointnet_bwd_scratch_floats(const act_pointnet_dims_t* d) { if (bad_pn(d)) return 0; PnBwdScratch sc; return carve_pn_bwd(nullptr, *d, sc); }

int act_pointnet_fwd_f32(const act_pointnet_dims_t* d, const act_pointnet_params_t* w, const float* x, int training, int keep_for_backward,
                         float* saved, float* out, float* ws, size_t wsb, act_stream_t stream) {
    return act_pointnet_fwd_groups_f32(d, w, x, training, keep_for_backward, saved, out, nullptr, 0, ws, wsb, stream);
}

int act_pointnet_fwd_groups_f32(const act_pointnet_dims_t* d, const act_pointnet_params_t* w, const float* x, int training, int keep_for_backward,
                                float* saved, float* out, const int32_t* groups, int n_groups, float* ws, size_t wsb, act_stream_t stream) {
    if (!w || !x || !saved || !out) return ACT_E_NULLPTR;
    if (bad_pn(d)) return ACT_E_BADARG;
    if (groups && (n_groups <= 0 || ((long long)n_groups * d->n) % 128)) return ACT_E_BADARG;
    if (groups && !pn_fused(*d)) groups = nullptr;                  // (the one-kernel-per-layer schedule computes every group)
    hipStream_t s = (hipStream_t)stream;
    const int BG = d->BG, n = d->n, C = d->C, R = BG * n;
    const bool fused = pn_fused(*d);
    PnSaved sv; carve_pn(saved, *d, sv);
    float *scale1 = sv.st1 + 2 * 128, *shift1 = sv.st1 + 3 * 128, *scale2 = sv.st2 + 2 * 512, *shift2 = sv.st2 + 3 * 512;
    // statistics of a tensor that is already in HBM (BatchNorm-1 always: its producer is the K = 3 conv; BatchNorm-2 on the plain schedule)
    auto stats = [&](const float* h, int Cc, const float* g_, const float* b_, float* rm, float* rv, float eps, float mom, float* st) -> int {
        if (training) {
            if (wsb < act_colstats_workspace(R, Cc)) return ACT_E_BADARG;
            RUN(act_bn_stats_f32(h, R, Cc, g_, b_, eps, mom, rm, rv, st, st + Cc, st + 2 * Cc, st + 3 * Cc, ws, wsb, s));
        } else {
            RUN(act_bn_eval_affine_f32(g_, b_, rm, rv, eps, Cc, st + 2 * Cc, st + 3 * Cc, s));
        }
        return 0;
    };
    act_gemm_epilogue_t e = epi0(); e.bias = w->c1_b;
    CK(gemm_nt(R, 128, 3, x, 3, w->c1_w, 3, sv.h1, 128, e, ws, wsb, s));
    CK(stats(sv.h1, 128, w->bn1_w, w->bn1_b, w->bn1_mean, w->bn1_var, d->eps1, d->momentum1, sv.st1));
    if (!fused) {
        RUN(act_affine_act_f32(sv.h1, scale1, shift1, 1, R, 128, sv.a1, s));
        e = epi0(); e.bias = w->c2_b;
        CK(gemm_nt(R, 256, 128, sv.a1, 128, w->c2_w, 128, sv.h2, 256, e, ws, wsb, s));
        RUN(act_group_max_f32(sv.h2, BG, n, 256, sv.fg, sv.arg1, s));
    } else {                                            // conv 128->256 on relu(bn1(h1)) applied on load; max over the group in the epilogue
        act_gemm_fx_t fx{}; fx.a_scale = scale1; fx.a_shift = shift1; fx.gmax = sv.fg; fx.garg = sv.arg1; fx.group = n; fx.store_c = 1;
        e = epi0(); e.bias = w->c2_b;
        CK(gemm_fx(1, 1, R, 256, 128, sv.h1, 128, w->c2_w, 128, sv.h2, 256, e, fx, ws, wsb, s));
    }
    // conv 512->512 on cat(global, local): the global half once per group, broadcast-added in the epilogue of the local half
    e = epi0(); e.bias = w->c3_b;
    CK(gemm_nt(BG, 512, 256, sv.fg, 256, w->c3_w, 512, sv.gw, 512, e, ws, wsb, s));
    e = epi0(); e.res = sv.gw; e.ldr = 512; e.res_row_div = n;
    if (!fused) {
        CK(gemm_nt(R, 512, 256, sv.h2, 256, w->c3_w + 256, 512, sv.h3, 512, e, ws, wsb, s));
        CK(stats(sv.h3, 512, w->bn2_w, w->bn2_b, w->bn2_mean, w->bn2_var, d->eps2, d->momentum2, sv.st2));
        RUN(act_affine_act_f32(sv.h3, scale2, shift2, 1, R, 512, sv.a3, s));
        e = epi0(); e.bias = w->c4_b;
        CK(gemm_nt(R, C, 512, sv.a3, 512, w->c4_w, 512, sv.h4, C, e, ws, wsb, s));
        RUN(act_group_max_f32(sv.h4, BG, n, C, out, keep_for_backward ? sv.arg2 : nullptr, s));
        return 0;
    }
    if (training) {                                     // BatchNorm-2 statistics from the epilogue of the conv that produces h3
        act_gemm_fx_t fx{}; fx.tile_stats = sv.tstats; fx.store_c = 1;
        CK(gemm_fx(1, 1, R, 512, 256, sv.h2, 256, w->c3_w + 256, 512, sv.h3, 512, e, fx, ws, wsb, s));
        RUN(act_bn_tiles_finalize_f32(sv.tstats, R / 128, 128, 512, w->bn2_w, w->bn2_b, d->eps2, d->momentum2, w->bn2_mean, w->bn2_var, sv.st2, sv.st2 + 512,
                                      scale2, shift2, s));
    } else {
        CK(gemm_nt(R, 512, 256, sv.h2, 256, w->c3_w + 256, 512, sv.h3, 512, e, ws, wsb, s));
        RUN(act_bn_eval_affine_f32(w->bn2_w, w->bn2_b, w->bn2_mean, w->bn2_var, d->eps2, 512, scale2, shift2, s));
    }
    {   // conv 512->C on relu(bn2(h3)) applied on load; only the max over the group leaves the kernel
        act_gemm_fx_t fx{}; fx.a_scale = scale2; fx.a_shift = shift2; fx.gmax = out; fx.garg = keep_for_backward ? sv.arg2 : nullptr; fx.group = n; fx.store_c = 0;
        e = epi0(); e.bias = w->c4_b;
        if (groups) {                                               // only the listed groups' tokens are wanted: the rest is a constant zero
            if (!t_collect) {
                if (hipMemsetAsync(out, 0, (size_t)BG * C * sizeof(float), s) != hipSuccess) return ACT_E_BADARG;
                // arg-max of an unlisted group = -1 (no row): every backward form drops an out-of-range arg (arg == r tests, (unsigned)arg < n),
                // so a gradient row that a caller hands in for an unlisted group reaches neither dX nor dW4 -- its true gradient is zero
                if (keep_for_backward && hipMemsetAsync(sv.arg2, 0xFF, (size_t)BG * C * sizeof(int32_t), s) != hipSuccess) return ACT_E_BADARG;
            }
            fx.row_groups = groups;
            CK(gemm_fx(1, 1, n_groups * n, C, 512, sv.h3, 512, w->c4_w, 512, nullptr, C, e, fx, ws, wsb, s));
        } else {
            CK(gemm_fx(1, 1, R, C, 512, sv.h3, 512, w->c4_w, 512, nullptr, C, e, fx, ws, wsb, s));
        }
    }
    return 0;
}

int act_pointnet_bwd_f32(const act_pointnet_dims_t* d, const act_pointnet_params_t* w, const float* x, const float* saved, const float* dout,
                         const act_pointnet_grads_t* g, float* scratch, float* ws, size_t wsb, act_stream_t stream) {
    if (!w || !x || !saved || !dout || !g || !scratch) return ACT_E_NULLPTR;
    if (bad_pn(d)) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int BG = d->BG, n = d->n, C = d->C, R = BG * n;
    const bool fused = pn_fused(*d);
    PnSaved sv; carve_pn(const_cast<float*>(saved), *d, sv);
    PnBwdScratch sc; carve_pn_bwd(scratch, *d, sc);
    auto st = [](float* base, int Cc, int which) { return base + which * Cc; };               // 0 mean, 1 rstd, 2 scale, 3 shift
    // dW = dY^T . relu(bn(h)) for a conv behind a BatchNorm: the activated operand is kept (plain) or recomputed while it is staged (fused)
    auto wgrad_act = [&](const float* dy, int N_, const float* h, const float* a, float* stt, int Cc, float* dw) -> int {
        if (!fused) return gemm_tn(N_, Cc, R, dy, N_, a, Cc, dw, Cc, ws, wsb, s);
        if (N_ % 128 == 0 && Cc % 128 == 0) {
            act_gemm_fx_t fx{}; fx.b_scale = st(stt, Cc, 2); fx.b_shift = st(stt, Cc, 3);
            return gemm_fx(0, 0, N_, Cc, R, dy, N_, h, Cc, dw, Cc, epi0(), fx, ws, wsb, s);
        }
        RUN(act_affine_act_f32(h, st(stt, Cc, 2), st(stt, Cc, 3), 1, R, Cc, sc.act, s));
        return gemm_tn(N_, Cc, R, dy, N_, sc.act, Cc, dw, Cc, ws, wsb, s);
    };
    const bool pool_on_load = pn_pool_bwd_on_load(*d);
    const int32_t* live = nullptr;
    if (pool_on_load && pn_pool_bwd_sparse()) {
        // dh4 has one live entry per (group, channel): both products walk those instead of a dense [R, C] operand (pool_bwd.hip)
        RUN(act_group_max_bwd_wgrad_f32(dout, sv.arg2, BG, n, C, sv.h3, 512, 512, st(sv.st2, 512, 2), st(sv.st2, 512, 3), g->c4_w, 512, ws, wsb, s));
        CK(colsum(dout, BG, C, g->c4_b, ws, wsb, s));
        // da3 of a group without gradient (a masked patch) is zero: not written here, not read by the BatchNorm backward below
        live = pn_pool_bwd_live() ? sc.live : nullptr;
        if (live) RUN(act_group_live_i32(dout, BG, C, sc.live, s));
        RUN(act_group_max_bwd_matmul_live_f32(dout, sv.arg2, BG, n, C, w->c4_w, 512, 512, sc.da3, 512, live, s));
    } else if (pool_on_load) {
        // dW4 = dh4^T . relu(bn2(h3)) and da3 = dh4 . W4 with dh4[r][c] = (arg2[r/n][c] == r % n ? dout[r/n][c] : 0) generated on load
        act_gemm_fx_t fx{}; fx.sa_src = dout; fx.sa_arg = sv.arg2; fx.group = n; fx.b_scale = st(sv.st2, 512, 2); fx.b_shift = st(sv.st2, 512, 3);
        CK(gemm_fx(0, 0, C, 512, R, nullptr, C, sv.h3, 512, g->c4_w, 512, epi0(), fx, ws, wsb, s));
        CK(colsum(dout, BG, C, g->c4_b, ws, wsb, s));                                         // sum_r dh4[r,c] = sum_g dout[g,c]
        act_gemm_fx_t fa{}; fa.sa_src = dout; fa.sa_arg = sv.arg2; fa.group = n;
        CK(gemm_fx(1, 0, R, 512, C, nullptr, C, w->c4_w, 512, sc.da3, 512, epi0(), fa, ws, wsb, s));
    } else {
        RUN(act_group_max_bwd_f32(dout, sv.arg2, BG, n, C, 0, sc.dh4, s));
        CK(wgrad_act(sc.dh4, C, sv.h3, sv.a3, sv.st2, 512, g->c4_w));
        if (fused) CK(colsum(dout, BG, C, g->c4_b, ws, wsb, s));                              // sum_r dh4[r,c] = sum_g dout[g,c]
        else       CK(colsum(sc.dh4, R, C, g->c4_b, ws, wsb, s));
        CK(gemm_nn(R, 512, C, sc.dh4, C, w->c4_w, 512, sc.da3, 512, epi0(), ws, wsb, s));
    }
    RUN(act_bn_bwd_groups_f32(sv.h3, sc.da3, st(sv.st2, 512, 2), st(sv.st2, 512, 3), st(sv.st2, 512, 0), st(sv.st2, 512, 1), 1, R, 512, live, n, sc.dh3,
                              g->bn2_w, g->bn2_b, ws, wsb, s));
    // the two column halves of dW3 [512, 512]: [:, :256] from the per-group path, [:, 256:] from the per-point path
    CK(gemm_tn(512, 256, R, sc.dh3, 512, sv.h2, 256, g->c3_w + 256, 512, ws, wsb, s));
    RUN(act_group_sum_f32(sc.dh3, BG, n, 512, sc.dgw, s));
    CK(gemm_tn(512, 256, BG, sc.dgw, 512, sv.fg, 256, g->c3_w, 512, ws, wsb, s));
    CK(colsum(sc.dgw, BG, 512, g->c3_b, ws, wsb, s));
    CK(gemm_nn(BG, 256, 512, sc.dgw, 512, w->c3_w, 512, sc.dfg, 256, epi0(), ws, wsb, s));
    if (pool_on_load) {                                 // dh2 = dh3 . W3[:, 256:] + the first pool's backward of dfg, added in the epilogue
        act_gemm_fx_t fe{}; fe.ep_src = sc.dfg; fe.ep_arg = sv.arg1; fe.group = n;
        CK(gemm_fx(1, 0, R, 256, 512, sc.dh3, 512, w->c3_w + 256, 512, sc.dh2, 256, epi0(), fe, ws, wsb, s));
    } else {
        CK(gemm_nn(R, 256, 512, sc.dh3, 512, w->c3_w + 256, 512, sc.dh2, 256, epi0(), ws, wsb, s));
        RUN(act_group_max_bwd_f32(sc.dfg, sv.arg1, BG, n, 256, 1, sc.dh2, s));
    }
    CK(wgrad_act(sc.dh2, 256, sv.h1, sv.a1, sv.st1, 128, g->c2_w));
    CK(colsum(sc.dh2, R, 256, g->c2_b, ws, wsb, s));
    CK(gemm_nn(R, 128, 256, sc.dh2, 256, w->c2_w, 128, sc.da1, 128, epi0(), ws, wsb, s));
    RUN(act_bn_bwd_f32(sv.h1, sc.da1, st(sv.st1, 128, 2), st(sv.st1, 128, 3), st(sv.st1, 128, 0), st(sv.st1, 128, 1), 1, R, 128, sc.dh1, g->bn1_w, g->bn1_b,
                       ws, wsb, s));
    CK(gemm_tn(128, 3, R, sc.dh1, 128, x, 3, g->c1_w, 3, ws, wsb, s));
    CK(colsum(sc.dh1, R, 128, g->c1_b, ws, wsb, s));
    return 0;
}

// ============================================================================================== DGCNN (inference form)
static const int kDgcnnCin[4] = {128, 256, 512, 512}, kDgcnnCout[4] = {256, 512, 512, 1024};
static size_t carve_dgcnn(float* base, const act_dgcnn_t& m, float*& x0, float*& yz, float*& cat, float*& stats) {
    const size_t T = (size_t)m.B * m.G;
    Carver c(base);
    x0 = c.take(T * 128); yz = c.take(T * 2048); cat = c.take(T * 2304); stats = c.take((size_t)18 * m.B * m.groups);
    return c.used;
}
size_t act_dgcnn_scratch_floats(const act_dgcnn_t* m) {
    if (!m || m->B <= 0 || m->G <= 0 || m->groups <= 0) return 0;
    float *a, *b, *c, *d; return carve_dgcnn(nullptr, *m, a, b, c, d);
}
int act_dgcnn_features_f32(const act_dgcnn_t* m, const float* f, const int64_t* idx, float* h, float* scratch, float* ws, size_t wsb,
                           act_stream_t stream) {
    if (!m || !f || !idx || !h || !scratch) return ACT_E_NULLPTR;
    if (m->B <= 0 || m->G <= 0 || m->k <= 0 || m->Cin <= 0 || m->Cout <= 0 || m->groups <= 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int T = m->B * m->G;
    float *x0, *yz, *cat, *stats;
    carve_dgcnn(scratch, *m, x0, yz, cat, stats);
    act_gemm_epilogue_t e = epi0(); e.bias = m->b_in;
    CK(gemm_nt(T, 128, m->Cin, f, m->Cin, m->w_in, m->Cin, x0, 128, e, ws, wsb, s));
    const float* xin = x0; int ldx = 128, off = 0;
    for (int l = 0; l < 4; ++l) {
        const int cin = kDgcnnCin[l], cout = kDgcnnCout[l];
        // W.cat(x_j - x_i, x_i) = Wa x_j + (Wb - Wa) x_i: one GEMM over the B*G points -> [Y | Z], then gather + GroupNorm + LeakyReLU + max_k
        CK(gemm_nt(T, 2 * cout, cin, xin, ldx, m->stacked[l], cin, yz, 2 * cout, epi0(), ws, wsb, s));
        RUN(act_edge_gn_lrelu_max_f32(yz, 2 * cout, cout, idx, m->B, m->G, m->k, cout, m->groups, m->gn_w[l], m->gn_b[l], m->eps, m->slope, stats,
                                     cat, 2304, off, s));
        xin = cat + off; ldx = 2304; off += cout;
    }
    CK(gemm_nt(T, m->Cout, 2304, cat, 2304, m->w5, 2304, h, m->Cout, epi0(), ws, wsb, s));
    return 0;
}

}  // extern "C"
