// point_ops.hip -- FPS, fused kNN-group, point gather, augmentation for gfx950 (CDNA4).
//
// Design (MI355X-first, not a translation of the CUDA wheels the reference calls):
//  * FPS is a chain of G-1 dependent arg-max steps, so the kernel minimises per-iteration
//    latency: every point of a cloud lives in VGPRs (contiguous chunk per lane), the running
//    min-distance never touches memory, the wave arg-max is a 6-step DPP reduction whose result
//    is broadcast through an SGPR (no LDS, no barrier when one wave holds the cloud), and the
//    winner's coordinates come from an LDS copy of the cloud read at a wave-uniform address.
//  * kNN-group: one wave per query; each lane owns a contiguous chunk of reference points as
//    fp32 distances in registers; K rounds of wave-wide extract-min (DPP min + ballot, lowest
//    lane == lowest index because chunks are contiguous) yield the ascending (dist, idx) list
//    directly; the same wave then gathers, centres and stores the neighbourhood.
//  * Distances use __fsub_rn/__fmul_rn/__fadd_rn: (dx*dx + dy*dy) + dz*dz, bit-identical to the
//    CPU oracle, so indices are bit-exact.
#include "common.h"

// four consecutive floats at a 4-BYTE aligned address: point k of cloud b sits at ref + (b * N + k) * 3 floats, which is 16-byte aligned only when
// (b * N + k) % 4 == 0 and ref itself is (N = 777, b = 1; an offset view).  gfx950 global loads have no alignment requirement, so this still compiles
// to one global_load_dwordx4 -- what changes is that the compiler may no longer ASSUME 16 bytes (round-3 advisor finding).
struct __attribute__((aligned(4))) float4_a4 { float x, y, z, w; };

#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

// =============================================== FPS ==============================================
// LDS_CLOUD is a template parameter on purpose: a runtime LDS-or-global choice compiles to FLAT loads, whose latency sits on
// the critical path of every one of the G-1 dependent iterations.
template <int WAVES, int PPL, bool LDS_CLOUD>
__global__ __launch_bounds__(WAVES * 64) void fps_kernel(const float* __restrict__ xyz, int N, int G,
                                                         int32_t* __restrict__ idx_out,
                                                         float* __restrict__ centers_out, int skip_near_origin) {
    constexpr bool lds_cloud = LDS_CLOUD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // layout: [2][WAVES] best value, [2][WAVES] best index, then (optional) the cloud xyz
    float* s_val = smem;
    int* s_idx = reinterpret_cast<int*>(smem + 2 * WAVES);
    float* s_xyz = smem + 4 * WAVES;

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    const int base = tid * PPL;

    float px[PPL], py[PPL], pz[PPL], td[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const int k = base + j;
        if (k < N) {
            px[j] = p[k * 3 + 0]; py[j] = p[k * 3 + 1]; pz[j] = p[k * 3 + 2];
            td[j] = 1e10f;
            if (skip_near_origin) {
                const float mag = __fadd_rn(__fadd_rn(__fmul_rn(px[j], px[j]), __fmul_rn(py[j], py[j])), __fmul_rn(pz[j], pz[j]));
                if (mag <= 1e-3f) td[j] = -1.0f;      // dead: min(d,-1) stays -1, never wins
            }
            if (lds_cloud) { s_xyz[k * 3 + 0] = px[j]; s_xyz[k * 3 + 1] = py[j]; s_xyz[k * 3 + 2] = pz[j]; }
        } else {
            px[j] = 0.f; py[j] = 0.f; pz[j] = 0.f; td[j] = -1.0f;
        }
    }
    if (WAVES > 1 || lds_cloud) __syncthreads();

    // the selected indices are collected in LDS and written once at the end: a global store inside the loop would put a
    // vmcnt(0) wait in front of every workgroup barrier and serialise each iteration with a memory round trip
    int* s_sel = reinterpret_cast<int*>(smem + 4 * WAVES) + (lds_cloud ? N * 3 : 0);
    int* s_vali = reinterpret_cast<int*>(s_val);
    int old = 0;
    float cx = p[0], cy = p[1], cz = p[2];
    if (tid == 0) s_sel[0] = 0;

    for (int it = 1; it < G; ++it) {
        // distances are >= 0 (dead / padding: -1): their bit patterns compare like signed integers, so the wave reduction
        // runs on integers and carries no float canonicalisation ops in its dependent chain
        int best = (int)0x80000000; int bj = 0;
        // (packed fp32 -- v_pk_add_f32 / v_pk_mul_f32 on point pairs, bit-exact -- was measured here and is NOT used: 551 vs 525 us for
        // 32 x 8192 -> 512, 1,116 vs 1,066 us for 128 x 8192 -> 1,024; the pairs cost register moves on the coordinates that stay resident
        // for the whole launch.  The kNN distance fill, which streams its points once, does gain from it: see knn_group2_kernel.)
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const float d = sqdist3(px[j], py[j], pz[j], cx, cy, cz);
            const float t = fminf(td[j], d);
            td[j] = t;
            const int ti = __float_as_int(t);
            if (ti > best) { best = ti; bj = j; }      // strict '>' : lowest j wins inside the lane
        }
        const int wmax = wave_max_i32(best, (int)0x80000000);
        const int wl = first_lane(__ballot(best == wmax));     // lowest lane == lowest index (contiguous chunks)
        int widx = __builtin_amdgcn_readlane(base + bj, wl);
        if (WAVES > 1) {
            const int par = it & 1;
            if (lane == 0) { s_vali[par * WAVES + wave] = wmax; s_idx[par * WAVES + wave] = widx; }
            __syncthreads();
            if constexpr (WAVES >= 8) {
                // 8 / 16 candidates: lane w of every wave takes wave w's (value, index) and the maximum is a 4-step DPP row reduction + ballot
                // (lowest lane = lowest wave = lowest index on ties) instead of a dependent scan of WAVES LDS entries by every lane
                const int v = lane < WAVES ? s_vali[par * WAVES + lane] : (int)0x80000000;
                const int id = lane < WAVES ? s_idx[par * WAVES + lane] : 0;
                int mx = v;
                mx = max(mx, dpp_mov_i<0x111, 0xf>((int)0x80000000, mx));
                mx = max(mx, dpp_mov_i<0x112, 0xf>((int)0x80000000, mx));
                mx = max(mx, dpp_mov_i<0x114, 0xf>((int)0x80000000, mx));
                mx = max(mx, dpp_mov_i<0x118, 0xf>((int)0x80000000, mx));
                const int gmax = __builtin_amdgcn_readlane(mx, 15);
                const int gl = first_lane(__ballot(v == gmax && lane < WAVES));
                widx = __builtin_amdgcn_readlane(id, gl);
            } else {
                int gv = s_vali[par * WAVES]; int gi = s_idx[par * WAVES];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) {
                    const int v = s_vali[par * WAVES + w];
                    if (v > gv) { gv = v; gi = s_idx[par * WAVES + w]; }
                }
                widx = gi;
            }
        }
        old = widx;
        if (lds_cloud) { cx = s_xyz[old * 3 + 0]; cy = s_xyz[old * 3 + 1]; cz = s_xyz[old * 3 + 2]; }
        else           { cx = p[old * 3 + 0];     cy = p[old * 3 + 1];     cz = p[old * 3 + 2]; }
        if (tid == 0) s_sel[it] = old;
    }
    __syncthreads();
    for (int g = tid; g < G; g += WAVES * 64) {
        const int k = s_sel[g];
        idx_out[(size_t)b * G + g] = k;
        if (centers_out) {
            float* c = centers_out + ((size_t)b * G + g) * 3;
            c[0] = p[k * 3 + 0]; c[1] = p[k * 3 + 1]; c[2] = p[k * 3 + 2];
        }
    }
}

// fallback for clouds too large for registers: running distances in global scratch-free LDS-less loop
__global__ __launch_bounds__(1024) void fps_big_kernel(const float* __restrict__ xyz, int N, int G,
                                                       int32_t* __restrict__ idx_out, float* __restrict__ centers_out,
                                                       float* __restrict__ temp, int skip_near_origin) {
    __shared__ float s_val[2][16];
    __shared__ int s_idx[2][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    float* __restrict__ t = temp + (size_t)b * N;
    const int per = (N + 1023) / 1024;                // contiguous chunk per thread keeps index order == thread order
    const int k0 = tid * per, k1 = min(N, k0 + per);
    for (int k = k0; k < k1; ++k) {
        float v = 1e10f;
        if (skip_near_origin) {
            const float x = p[k * 3], y = p[k * 3 + 1], z = p[k * 3 + 2];
            if (__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)) <= 1e-3f) v = -1.0f;
        }
        t[k] = v;
    }
    float cx = p[0], cy = p[1], cz = p[2];
    if (tid == 0) { idx_out[(size_t)b * G] = 0; if (centers_out) { float* c = centers_out + (size_t)b * G * 3; c[0] = cx; c[1] = cy; c[2] = cz; } }
    for (int it = 1; it < G; ++it) {
        float best = -2.0f; int bi = 0;
        for (int k = k0; k < k1; ++k) {
            const float d = sqdist3(p[k * 3], p[k * 3 + 1], p[k * 3 + 2], cx, cy, cz);
            const float v = fminf(t[k], d);
            t[k] = v;
            if (v > best) { best = v; bi = k; }
        }
        const float wmax = wave_max_f32(best, -3.0f);
        const int wl = first_lane(__ballot(best == wmax));
        const int widx = __builtin_amdgcn_readlane(bi, wl);
        const int par = it & 1;
        if (lane == 0) { s_val[par][wave] = wmax; s_idx[par][wave] = widx; }
        __syncthreads();
        float gv = s_val[par][0]; int gi = s_idx[par][0];
        for (int w = 1; w < 16; ++w) { const float v = s_val[par][w]; if (v > gv) { gv = v; gi = s_idx[par][w]; } }
        cx = p[gi * 3]; cy = p[gi * 3 + 1]; cz = p[gi * 3 + 2];
        if (tid == 0) { idx_out[(size_t)b * G + it] = gi; if (centers_out) { float* c = centers_out + ((size_t)b * G + it) * 3; c[0] = cx; c[1] = cy; c[2] = cz; } }
    }
}

template <int WAVES, int PPL>
static int launch_fps(const float* xyz, int B, int N, int G, int32_t* idx, float* centers, int skip, hipStream_t s) {
    const size_t cloud_bytes = (size_t)N * 3 * sizeof(float);
    const int lds_cloud = cloud_bytes <= 128 * 1024 ? 1 : 0;
    const size_t smem = 4 * WAVES * sizeof(float) + (lds_cloud ? cloud_bytes : 0) + (size_t)G * sizeof(int);
    if (smem > 160 * 1024) return ACT_E_BADARG;
    auto k = lds_cloud ? fps_kernel<WAVES, PPL, true> : fps_kernel<WAVES, PPL, false>;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k, dim3(B), dim3(WAVES * 64), smem, s, xyz, N, G, idx, centers, skip);
    ACT_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t act_fps_scratch_floats(int B, int N) { return N > 16384 ? (size_t)B * N : 0; }

extern "C" int act_fps_f32(const float* xyz, int B, int N, int G, int32_t* idx_out, float* centers_out,
                           int skip_near_origin, float* scratch, act_stream_t stream) {
    if (B == 0 || G == 0) return 0;                        // empty batch: nothing to do (empty tensors have NULL storage)
    if (!xyz || !idx_out) return ACT_E_NULLPTR;
    if (B < 0 || N <= 0 || G < 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    // algorithmic bytes: read xyz once, write idx (+ centers)  [SURVEY 8d]
    ActProfScope ps(KID_FPS, s, 0.0, (double)B * (12.0 * N + 4.0 * G + (centers_out ? 12.0 * G : 0.0)));
    { const char* e = getenv("ACT_FPS_CFG");          // tuning knob for N <= 1024: 1 = one wave x 16 points/lane, 2 = two waves x 8
      if (e && N <= 1024 && N > 256) { if (atoi(e) == 1) return launch_fps<1, 16>(xyz, B, N, G, idx_out, centers_out, skip_near_origin, s);
                                       if (atoi(e) == 2) return launch_fps<2, 8>(xyz, B, N, G, idx_out, centers_out, skip_near_origin, s);
                                       if (atoi(e) == 3) return launch_fps<8, 2>(xyz, B, N, G, idx_out, centers_out, skip_near_origin, s);
                                       if (atoi(e) == 4) return launch_fps<16, 1>(xyz, B, N, G, idx_out, centers_out, skip_near_origin, s); } }
    if (N <= 256)   return launch_fps<1, 4>(xyz, B, N, G, idx_out, centers_out, skip_near_origin, s);
    if (N <= 1024)  return launch_fps<4, 4>(xyz, B, N, G, idx_out, centers_out, skip_near_origin, s);
    if (N <= 2048)  return launch_fps<4, 8>(xyz, B, N, G, idx_out, centers_out, skip_near_origin, s);
    if (N <= 4096)  return launch_fps<8, 8>(xyz, B, N, G, idx_out, centers_out, skip_near_origin, s);
    if (N <= 8192)  return launch_fps<16, 8>(xyz, B, N, G, idx_out, centers_out, skip_near_origin, s);
    if (N <= 16384) return launch_fps<16, 16>(xyz, B, N, G, idx_out, centers_out, skip_near_origin, s);
    // very large clouds: the running distances live in the caller's scratch (act_fps_scratch_floats floats)
    if (!scratch) return ACT_E_NULLPTR;
    hipLaunchKernelGGL(fps_big_kernel, dim3(B), dim3(1024), 0, s, xyz, N, G, idx_out, centers_out, scratch, skip_near_origin);
    ACT_LAUNCH_CHECK();
    return 0;
}

// ---- latency probes of the FPS critical path (measurement infrastructure for bench.py `group_fps_knn.fps_chain`; not on any product path) -------------
// One FPS iteration is a chain of four dependent phases; each phase is timed on its own as `iters` dependent repetitions inside ONE workgroup (the chip is
// otherwise idle, so this is latency, not throughput), with s_memrealtime (100 MHz, constant) stamps taken by thread 0:
//   0 eval    PPL x (sqdist3 + min + integer compare / select) on register-resident points     -- VALU issue of one wave (every wave does the same in parallel)
//   1 argmax  6-step DPP integer max + ballot + readlane                                        -- the wave reduction
//   2 xwave   LDS slot write, s_barrier, the cross-wave arg-max (scan for WAVES < 8, DPP row reduction from 8 waves on)
//   3 centre  3 dependent LDS reads at a wave-uniform, data-dependent address                   -- the winner's coordinates
//   4 all     the four phases chained as in fps_kernel (a workgroup alone on the chip)
// The sum of 0..3 is the floor of a design that serialises the phases; `all` shows what the hardware overlaps between them.
template <int WAVES, int PPL>
__global__ __launch_bounds__(WAVES * 64) void fps_chain_probe_kernel(int iters, unsigned long long* __restrict__ ticks, int* __restrict__ sink) {
    __shared__ int s_vali[2 * WAVES], s_idx[2 * WAVES];
    __shared__ float s_xyz[WAVES * 64 * PPL * 3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, base = tid * PPL;
    constexpr int N = WAVES * 64 * PPL;
    float px[PPL], py[PPL], pz[PPL], td[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const int k = base + j;
        px[j] = __sinf(0.37f * k); py[j] = __cosf(0.11f * k + 1.f); pz[j] = __sinf(0.05f * k + 2.f); td[j] = 1e10f;
        s_xyz[k * 3 + 0] = px[j]; s_xyz[k * 3 + 1] = py[j]; s_xyz[k * 3 + 2] = pz[j];
    }
    __syncthreads();
    float cx = px[0], cy = py[0], cz = pz[0];
    int best = 0, bj = 0, old = 0, acc = 0;
    unsigned long long t[6];
    auto eval = [&]() {
        best = (int)0x80000000; bj = 0;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const float d = sqdist3(px[j], py[j], pz[j], cx, cy, cz);
            const float tt = fminf(td[j], d);
            td[j] = tt;
            const int ti = __float_as_int(tt);
            if (ti > best) { best = ti; bj = j; }
        }
    };
    auto argmax = [&]() -> int {
        const int wmax = wave_max_i32(best, (int)0x80000000);
        const int wl = first_lane(__ballot(best == wmax));
        const int widx = __builtin_amdgcn_readlane(base + bj, wl);
        if (lane == 0) { s_vali[wave] = wmax; s_idx[wave] = widx; }      // (kept out of the timed dependency of phase 1 by the caller)
        return widx;
    };
    auto xwave = [&](int par, int wmax, int widx) -> int {
        if (WAVES == 1) return widx;
        if (lane == 0) { s_vali[par * WAVES + wave] = wmax; s_idx[par * WAVES + wave] = widx; }
        __syncthreads();
        if constexpr (WAVES >= 8) {
            const int v = lane < WAVES ? s_vali[par * WAVES + lane] : (int)0x80000000;
            const int id = lane < WAVES ? s_idx[par * WAVES + lane] : 0;
            int mx = v;
            mx = max(mx, dpp_mov_i<0x111, 0xf>((int)0x80000000, mx));
            mx = max(mx, dpp_mov_i<0x112, 0xf>((int)0x80000000, mx));
            mx = max(mx, dpp_mov_i<0x114, 0xf>((int)0x80000000, mx));
            mx = max(mx, dpp_mov_i<0x118, 0xf>((int)0x80000000, mx));
            const int gmax = __builtin_amdgcn_readlane(mx, 15);
            const int gl = first_lane(__ballot(v == gmax && lane < WAVES));
            return __builtin_amdgcn_readlane(id, gl);
        } else {
            int gv = s_vali[par * WAVES]; int gi = s_idx[par * WAVES];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) {
                const int v = s_vali[par * WAVES + w];
                if (v > gv) { gv = v; gi = s_idx[par * WAVES + w]; }
            }
            return gi;
        }
    };
    // ---- phase 0: distance evaluations (the chain runs through cx: the next iteration's centre depends on this one's result)
    __syncthreads(); t[0] = wall_clock64();
    for (int it = 0; it < iters; ++it) { eval(); cx = __int_as_float(__float_as_int(cx) ^ (best & 1)); }
    acc += best + bj;
    // ---- phase 1: wave arg-max
    __syncthreads(); t[1] = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        const int wmax = wave_max_i32(best, (int)0x80000000);
        const int wl = first_lane(__ballot(best == wmax));
        const int widx = __builtin_amdgcn_readlane(base + bj, wl);
        best = best ^ (widx & 1); bj = (bj + widx) & (PPL - 1);
    }
    acc += best;
    // ---- phase 2: cross-wave arg-max through LDS + barrier
    __syncthreads(); t[2] = wall_clock64();
    { int w = best, id = base;
      for (int it = 0; it < iters; ++it) { id = xwave(it & 1, w, id); w ^= id & 1; }
      acc += id + w; }
    // ---- phase 3: the winner's coordinates from LDS
    __syncthreads(); t[3] = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        cx = s_xyz[old * 3 + 0]; cy = s_xyz[old * 3 + 1]; cz = s_xyz[old * 3 + 2];
        old = (__float_as_int(cx) ^ __float_as_int(cy) ^ __float_as_int(cz)) & (N - 1);
    }
    acc += old;
    // ---- phase 4: the whole iteration, as in fps_kernel
    __syncthreads(); t[4] = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        eval();
        const int wmax = wave_max_i32(best, (int)0x80000000);
        const int wl = first_lane(__ballot(best == wmax));
        int widx = __builtin_amdgcn_readlane(base + bj, wl);
        widx = xwave(it & 1, wmax, widx);
        old = widx & (N - 1);
        cx = s_xyz[old * 3 + 0]; cy = s_xyz[old * 3 + 1]; cz = s_xyz[old * 3 + 2];
    }
    acc += old + best;
    __syncthreads(); t[5] = wall_clock64();
    if (tid == 0) for (int i = 0; i < 5; ++i) ticks[i] = t[i + 1] - t[i];
    if (acc == 0x7fffffff) sink[tid] = acc;                                 // keep every chain alive
}

// us_per_iteration[5] = {eval, argmax, xwave, centre, all} for the launch configuration act_fps_f32 uses at this N (N <= 8192); synchronises the stream
extern "C" int act_fps_chain_probe(int N, int iters, double* us_per_iteration, int* waves_out, int* ppl_out, act_stream_t stream) {
    if (!us_per_iteration) return ACT_E_NULLPTR;
    if (N <= 0 || N > 8192 || iters <= 0) return ACT_E_BADARG;           // (larger clouds do not keep their coordinates in LDS)
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* ticks = nullptr; int* sink = nullptr;
    if (hipMalloc(&ticks, 5 * sizeof(unsigned long long)) != hipSuccess) return ACT_E_BADARG;
    if (hipMalloc(&sink, 1024 * sizeof(int)) != hipSuccess) { (void)hipFree(ticks); return ACT_E_BADARG; }
    int W = 0, P = 0;
#define PROBE(W_, P_) { W = W_; P = P_; hipLaunchKernelGGL((fps_chain_probe_kernel<W_, P_>), dim3(1), dim3(W_ * 64), 0, s, iters, ticks, sink); }
    if (N <= 256) PROBE(1, 4) else if (N <= 1024) PROBE(4, 4) else if (N <= 2048) PROBE(4, 8) else if (N <= 4096) PROBE(8, 8) else PROBE(16, 8)
#undef PROBE
    unsigned long long h[5] = {0, 0, 0, 0, 0};
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, ticks, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(ticks); (void)hipFree(sink);
    if (e != hipSuccess) return (int)e;
    for (int i = 0; i < 5; ++i) us_per_iteration[i] = (double)h[i] * 0.01 / (double)iters;      // s_memrealtime: 100 MHz
    if (waves_out) *waves_out = W;
    if (ppl_out) *ppl_out = P;
    return 0;
}

// =========================================== kNN + group ==========================================
// one wave per query; PPL reference points per lane (contiguous), N <= 64*PPL
template <int PPL>
__global__ __launch_bounds__(256) void knn_group_kernel(const float* __restrict__ ref, const float* __restrict__ query,
                                                        int B, int N, int Q, int K, int64_t* __restrict__ idx_out,
                                                        int idx_kq, float* __restrict__ nbr_out,
                                                        float* __restrict__ dist_out) {
    const int lane = threadIdx.x & 63;
    const long long qid = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // b*Q + q
    if (qid >= (long long)B * Q) return;                 // whole wave exits together
    const int b = (int)(qid / Q), q = (int)(qid % Q);
    const float* __restrict__ r = ref + (size_t)b * N * 3;
    const float qx = query[qid * 3 + 0], qy = query[qid * 3 + 1], qz = query[qid * 3 + 2];
    const float INF = __int_as_float(0x7f800000);
    const int base = lane * PPL;

    float d[PPL];
    float lmin = INF; int lj = 0;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const int k = base + j;
        float v = INF;
        if (k < N) v = sqdist3(r[k * 3 + 0], r[k * 3 + 1], r[k * 3 + 2], qx, qy, qz);
        d[j] = v;
        if (v < lmin) { lmin = v; lj = j; }
    }
    int my_idx = 0; float my_d = 0.f;
    for (int round = 0; round < K; ++round) {
        const float m = wave_min_f32(lmin, INF);
        const int wl = first_lane(__ballot(lmin == m));
        const int widx = __builtin_amdgcn_readlane(base + lj, wl);
        if (lane == round) { my_idx = widx; my_d = m; }
        if (lane == wl) {                                // retire the winner, refresh this lane's minimum
            float nm = INF; int nj = 0;
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const float v = (j == lj) ? INF : d[j];
                d[j] = v;
                if (v < nm) { nm = v; nj = j; }
            }
            lmin = nm; lj = nj;
        }
    }
    if (lane < K) {
        const size_t o = idx_kq ? ((size_t)b * K + lane) * Q + q : (size_t)qid * K + lane;
        idx_out[o] = (int64_t)my_idx;
        if (dist_out) dist_out[o] = sqrtf(my_d);
        if (nbr_out) {
            float* __restrict__ w = nbr_out + ((size_t)qid * K + lane) * 3;
            w[0] = __fsub_rn(r[my_idx * 3 + 0], qx);
            w[1] = __fsub_rn(r[my_idx * 3 + 1], qy);
            w[2] = __fsub_rn(r[my_idx * 3 + 2], qz);
        }
    }
}


// PPL >= 64 (N up to 8192: the stress geometry, 128 points per lane): the flat refresh above rescans the winner lane's whole chunk in every
// one of the K rounds (K x PPL x ~3.5 instructions = 95 % of the kernel).  Two levels: the chunk is NG groups of 8 with cached group minima;
// a round rescans only the winner's group and folds the NG minima.  Only the winner lane is active in the refresh, so the statically unrolled
// `if (g == gsel)` chain executes exactly one group body (the others are skipped by execz branches) and every register index stays static.
// Selection order is unchanged: lowest distance, then lowest lane, then lowest index in the lane (strict '<' scanning upwards at both levels).
template <int PPL, int GS = 8>
__global__ __launch_bounds__(256) void knn_group2_kernel(const float* __restrict__ ref, const float* __restrict__ query,
                                                         int B, int N, int Q, int K, int64_t* __restrict__ idx_out,
                                                         int idx_kq, float* __restrict__ nbr_out,
                                                         float* __restrict__ dist_out) {
    constexpr int NG = PPL / GS;
    const int lane = threadIdx.x & 63;
    const long long qid = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // b*Q + q
    if (qid >= (long long)B * Q) return;
    const int b = (int)(qid / Q), q = (int)(qid % Q);
    const float* __restrict__ r = ref + (size_t)b * N * 3;
    const float qx = query[qid * 3 + 0], qy = query[qid * 3 + 1], qz = query[qid * 3 + 2];
    const float INF = __int_as_float(0x7f800000);
    // point layout: register element e of lane l is point  (e / 4) * 256 + 4 * l + (e % 4)  -- a lane owns 4 consecutive points (48 contiguous,
    // 16-byte aligned bytes = three float4 loads) and the 64 lanes of one load instruction sweep 3 KB of the cloud front to back, instead of
    // 64 private chunks 12 * PPL bytes apart (one cache line per lane and instruction: at 128 points per lane the fill, not the K selection
    // rounds, was the larger half of the kernel).  Inside a lane ascending e is still ascending point index; ACROSS lanes it no longer is,
    // so an exact distance tie between lanes is resolved by a second wave reduction on the point index (lowest index wins, as before).
    const int base = 4 * lane;
#define KNN_IDX(e) (((e) >> 2) * 256 + base + ((e) & 3))

    float d[PPL];
    float gmin[NG]; int gj[NG];                          // minimum of group g and its position inside the group
    // fill: four points = three float4 loads
    //   f0 = (x0 y0 z0 x1)  f1 = (y1 z1 x2 y2)  f2 = (z2 x3 y3 z3)
    // and the six component pairs of two points sit in those registers exactly as packed fp32 wants them: three v_pk_add_f32 + three
    // v_pk_mul_f32 give the six squared differences of a point pair (each half rounded like the scalar op), four scalar adds finish
    // (dx*dx + dy*dy) + dz*dz per point -- 5 instead of 8 VALU issues and 0.75 instead of 3 load instructions a point, same bits as the oracle
    static_assert(GS % 4 == 0, "groups of 4 or 8 points");
    const f32x2 qxy = {qx, qy}, qzx = {qz, qx}, qyz = {qy, qz};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float m = INF; int mj = 0;
#pragma unroll
        for (int i = 0; i < GS; i += 4) {
            const int k = KNN_IDX(g * GS + i);
            float v[4] = {INF, INF, INF, INF};
            if (k + 3 < N) {
                const float4_a4* __restrict__ pp = reinterpret_cast<const float4_a4*>(r + (size_t)k * 3);    // 4-byte aligned only: see float4_a4
                const float4_a4 f0 = pp[0], f1 = pp[1], f2 = pp[2];
                const f32x2 a = f32x2{f0.x, f0.y} - qxy, bq = f32x2{f0.z, f0.w} - qzx, c = f32x2{f1.x, f1.y} - qyz;
                const f32x2 e = f32x2{f1.z, f1.w} - qxy, f = f32x2{f2.x, f2.y} - qzx, h = f32x2{f2.z, f2.w} - qyz;
                const f32x2 a2 = a * a, b2 = bq * bq, c2 = c * c, e2 = e * e, f2s = f * f, h2 = h * h;
                v[0] = __fadd_rn(__fadd_rn(a2[0], a2[1]), b2[0]);
                v[1] = __fadd_rn(__fadd_rn(b2[1], c2[0]), c2[1]);
                v[2] = __fadd_rn(__fadd_rn(e2[0], e2[1]), f2s[0]);
                v[3] = __fadd_rn(__fadd_rn(f2s[1], h2[0]), h2[1]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k + u < N) v[u] = sqdist3(r[(k + u) * 3 + 0], r[(k + u) * 3 + 1], r[(k + u) * 3 + 2], qx, qy, qz);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                d[g * GS + i + u] = v[u];
                if (v[u] < m) { m = v[u]; mj = i + u; }
            }
        }
        gmin[g] = m; gj[g] = mj;
    }
    float lmin = INF; int lg = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) if (gmin[g] < lmin) { lmin = gmin[g]; lg = g; }
    int my_idx = 0; float my_d = 0.f;
    for (int round = 0; round < K; ++round) {
        const float m = wave_min_f32(lmin, INF);
        const unsigned long long tie = __ballot(lmin == m);
        int wl = first_lane(tie);
        int lj = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) if (g == lg) lj = g * GS + gj[g];
        const int lidx = KNN_IDX(lj);
        int widx = __builtin_amdgcn_readlane(lidx, wl);
        if (__builtin_popcountll(tie) > 1) {             // equal distances in several lanes (wave-uniform, rare): the lowest POINT INDEX wins
            widx = -wave_max_i32((lmin == m) ? -lidx : (int)0x80000001, (int)0x80000001);
            wl = first_lane(__ballot(lmin == m && lidx == widx));
        }
        if (lane == round) { my_idx = widx; my_d = m; }
        if (lane == wl) {                                // retire the winner: rescan its group, fold the group minima
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g == lg) {
                    float nm = INF; int nj = 0;
#pragma unroll
                    for (int i = 0; i < GS; ++i) {
                        const float v = (i == gj[g]) ? INF : d[g * GS + i];
                        d[g * GS + i] = v;
                        if (v < nm) { nm = v; nj = i; }
                    }
                    gmin[g] = nm; gj[g] = nj;
                }
            }
            float nl = INF; int ng = 0;
#pragma unroll
            for (int g = 0; g < NG; ++g) if (gmin[g] < nl) { nl = gmin[g]; ng = g; }
            lmin = nl; lg = ng;
        }
    }
    if (lane < K) {
        const size_t o = idx_kq ? ((size_t)b * K + lane) * Q + q : (size_t)qid * K + lane;
        idx_out[o] = (int64_t)my_idx;
        if (dist_out) dist_out[o] = sqrtf(my_d);
        if (nbr_out) {
            float* __restrict__ w = nbr_out + ((size_t)qid * K + lane) * 3;
            w[0] = __fsub_rn(r[my_idx * 3 + 0], qx);
            w[1] = __fsub_rn(r[my_idx * 3 + 1], qy);
            w[2] = __fsub_rn(r[my_idx * 3 + 2], qz);
        }
    }
}

#undef KNN_IDX

// ---- the same selection with the lane's distances kept in a 4-ary TOURNAMENT TREE of registers --------------------------------------------
// Retiring a winner in the two-level kernel above walks PPL / GS predicated group bodies, rescans one group and folds PPL / GS group minima:
// a dependent chain of ~100 instructions per round at 128 points per lane, on 2 waves per SIMD -- latency, not throughput.  Here every node of
// a fan-out-4 tree over the lane's points keeps (minimum, leaf offset of the minimum); retiring the current minimum descends ONE path
// (4 predicated bodies per level), rescans 4 leaves and re-folds 4 children per level on the way back: 3 levels for 64 points, a top node
// over 2 such trees for 128.  Same point layout, same selection order (strict '<' scanning upwards at every level; lane ties by point index).
// (flat per-level register arrays with compile-time indices after unrolling: a struct-of-structs formulation of the same tree was left in
// scratch memory by the compiler at 64 / 128 points per lane and ran 2x slower)
// fill the 4 leaves d[0..3] of one level-1 node: the lane's 4 consecutive points starting at point k
__device__ __forceinline__ void knn_fill4(float* __restrict__ d, int k, const float* __restrict__ r, int N, float qx, float qy, float qz) {
    const float INF = __int_as_float(0x7f800000);
    float v[4] = {INF, INF, INF, INF};
    if (k + 3 < N) {
        const float4_a4* __restrict__ pp = reinterpret_cast<const float4_a4*>(r + (size_t)k * 3);    // 4-byte aligned only: see float4_a4
        const float4_a4 f0 = pp[0], f1 = pp[1], f2 = pp[2];
        const f32x2 qxy = {qx, qy}, qzx = {qz, qx}, qyz = {qy, qz};
        const f32x2 a = f32x2{f0.x, f0.y} - qxy, bq = f32x2{f0.z, f0.w} - qzx, c = f32x2{f1.x, f1.y} - qyz;
        const f32x2 e = f32x2{f1.z, f1.w} - qxy, f = f32x2{f2.x, f2.y} - qzx, h = f32x2{f2.z, f2.w} - qyz;
        const f32x2 a2 = a * a, b2 = bq * bq, c2 = c * c, e2 = e * e, f2s = f * f, h2 = h * h;
        v[0] = __fadd_rn(__fadd_rn(a2[0], a2[1]), b2[0]);
        v[1] = __fadd_rn(__fadd_rn(b2[1], c2[0]), c2[1]);
        v[2] = __fadd_rn(__fadd_rn(e2[0], e2[1]), f2s[0]);
        v[3] = __fadd_rn(__fadd_rn(f2s[1], h2[0]), h2[1]);
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k + u < N) v[u] = sqdist3(r[(k + u) * 3 + 0], r[(k + u) * 3 + 1], r[(k + u) * 3 + 2], qx, qy, qz);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) d[u] = v[u];
}
// (minimum, offset of the minimum) of CNT children whose own (min, offset) are cm[i], ca[i]; child i covers SPAN leaves
#define KNN_FOLD(CNT, SPAN, cm, ca, out_m, out_a) do { float m__ = (cm)[0]; int a__ = (ca)[0]; \
    _Pragma("unroll") for (int i__ = 1; i__ < (CNT); ++i__) if ((cm)[i__] < m__) { m__ = (cm)[i__]; a__ = i__ * (SPAN) + (ca)[i__]; } \
    (out_m) = m__; (out_a) = a__; } while (0)

template <int PPL>
__global__ __launch_bounds__(256) void knn_tree_kernel(const float* __restrict__ ref, const float* __restrict__ query,
                                                       int B, int N, int Q, int K, int64_t* __restrict__ idx_out,
                                                       int idx_kq, float* __restrict__ nbr_out,
                                                       float* __restrict__ dist_out) {
    // levels: N1 nodes over 4 leaves; N2 nodes over 4 level-1 nodes (PPL >= 32); N3 nodes over 4 level-2 nodes (PPL = 128); the top folds
    // the nodes of the highest level (2 or 4 of them)
    constexpr int N1 = PPL / 4, N2 = PPL >= 32 ? PPL / 16 : 0, N3 = PPL >= 128 ? PPL / 64 : 0;
    static_assert(PPL == 8 || PPL == 16 || PPL == 32 || PPL == 64 || PPL == 128, "points per lane");
    const int lane = threadIdx.x & 63;
    const long long qid = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // b*Q + q
    if (qid >= (long long)B * Q) return;
    const int b = (int)(qid / Q), q = (int)(qid % Q);
    const float* __restrict__ r = ref + (size_t)b * N * 3;
    const float qx = query[qid * 3 + 0], qy = query[qid * 3 + 1], qz = query[qid * 3 + 2];
    const float INF = __int_as_float(0x7f800000);
    const int base = 4 * lane;

    float d[PPL];
    float m1[N1]; int a1[N1];
    float m2[N2 > 0 ? N2 : 1]; int a2[N2 > 0 ? N2 : 1];
    float m3[N3 > 0 ? N3 : 1]; int a3[N3 > 0 ? N3 : 1];
#pragma unroll
    for (int n = 0; n < N1; ++n) {
        knn_fill4(d + 4 * n, n * 256 + base, r, N, qx, qy, qz);
        float lm = d[4 * n]; int la = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i) if (d[4 * n + i] < lm) { lm = d[4 * n + i]; la = i; }
        m1[n] = lm; a1[n] = la;
    }
    if constexpr (N2 > 0) {
#pragma unroll
        for (int n = 0; n < N2; ++n) KNN_FOLD(4, 4, m1 + 4 * n, a1 + 4 * n, m2[n], a2[n]);
    }
    if constexpr (N3 > 0) {
#pragma unroll
        for (int n = 0; n < N3; ++n) KNN_FOLD(4, 16, m2 + 4 * n, a2 + 4 * n, m3[n], a3[n]);
    }
    float lmin; int lat;                                                 // the lane's minimum and its register element
    auto fold_top = [&]() {
        if constexpr (N3 > 0)      KNN_FOLD(N3, 64, m3, a3, lmin, lat);
        else if constexpr (N2 > 0) KNN_FOLD(N2, 16, m2, a2, lmin, lat);
        else                       KNN_FOLD(N1, 4, m1, a1, lmin, lat);
    };
    fold_top();
    int my_idx = 0; float my_d = 0.f;
    for (int round = 0; round < K; ++round) {
        const float m = wave_min_f32(lmin, INF);
        const unsigned long long tie = __ballot(lmin == m);
        int wl = first_lane(tie);
        const int lidx = (lat >> 2) * 256 + base + (lat & 3);
        int widx = __builtin_amdgcn_readlane(lidx, wl);
        if (__builtin_popcountll(tie) > 1) {             // equal distances in several lanes (wave-uniform, rare): the lowest POINT INDEX wins
            widx = -wave_max_i32((lmin == m) ? -lidx : (int)0x80000001, (int)0x80000001);
            wl = first_lane(__ballot(lmin == m && lidx == widx));
        }
        if (lane == round) { my_idx = widx; my_d = m; }
        if (lane == wl) {                                // retire leaf `lat`: one path down, 4 leaves rescanned, 4 children re-folded per level
            const int leaf = lat & 3;
            auto rescan = [&](int n) {                  // level-1 node n (a compile-time constant after unrolling): drop the leaf, new minimum of 4
                float lm = INF; int la = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = (i == leaf) ? INF : d[4 * n + i];
                    d[4 * n + i] = v;
                    if (v < lm) { lm = v; la = i; }
                }
                m1[n] = lm; a1[n] = la;
            };
            if constexpr (N3 > 0) {                     // 128 points: 2 + 4 + 4 predicated bodies on the way down
                const int w3 = lat >> 6, w2 = (lat >> 4) & 3, w1 = (lat >> 2) & 3;
#pragma unroll
                for (int n3 = 0; n3 < N3; ++n3) {
                    if (n3 == w3) {
#pragma unroll
                        for (int c2 = 0; c2 < 4; ++c2) {
                            if (c2 == w2) {
#pragma unroll
                                for (int c1 = 0; c1 < 4; ++c1) if (c1 == w1) rescan((n3 * 4 + c2) * 4 + c1);
                                KNN_FOLD(4, 4, m1 + 4 * (n3 * 4 + c2), a1 + 4 * (n3 * 4 + c2), m2[n3 * 4 + c2], a2[n3 * 4 + c2]);
                            }
                        }
                        KNN_FOLD(4, 16, m2 + 4 * n3, a2 + 4 * n3, m3[n3], a3[n3]);
                    }
                }
            } else if constexpr (N2 > 0) {              // 32 / 64 points: N2 + 4 predicated bodies
                const int w2 = lat >> 4, w1 = (lat >> 2) & 3;
#pragma unroll
                for (int n2 = 0; n2 < N2; ++n2) {
                    if (n2 == w2) {
#pragma unroll
                        for (int c1 = 0; c1 < 4; ++c1) if (c1 == w1) rescan(n2 * 4 + c1);
                        KNN_FOLD(4, 4, m1 + 4 * n2, a1 + 4 * n2, m2[n2], a2[n2]);
                    }
                }
            } else {
                const int w1 = lat >> 2;
#pragma unroll
                for (int n1 = 0; n1 < N1; ++n1) if (n1 == w1) rescan(n1);
            }
            fold_top();
        }
    }
    if (lane < K) {
        const size_t o = idx_kq ? ((size_t)b * K + lane) * Q + q : (size_t)qid * K + lane;
        idx_out[o] = (int64_t)my_idx;
        if (dist_out) dist_out[o] = sqrtf(my_d);
        if (nbr_out) {
            float* __restrict__ w = nbr_out + ((size_t)qid * K + lane) * 3;
            w[0] = __fsub_rn(r[my_idx * 3 + 0], qx);
            w[1] = __fsub_rn(r[my_idx * 3 + 1], qy);
            w[2] = __fsub_rn(r[my_idx * 3 + 2], qz);
        }
    }
}
#undef KNN_FOLD

// fallback for N > 8192: nothing is kept in registers; each of the K rounds rescans the lane's chunk for the
// smallest (distance, index) pair lexicographically greater than the previous winner.
__global__ __launch_bounds__(256) void knn_group_big_kernel(const float* __restrict__ ref, const float* __restrict__ query,
                                                            int B, int N, int Q, int K, int64_t* __restrict__ idx_out,
                                                            int idx_kq, float* __restrict__ nbr_out,
                                                            float* __restrict__ dist_out) {
    const int lane = threadIdx.x & 63;
    const long long qid = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (qid >= (long long)B * Q) return;
    const int b = (int)(qid / Q), q = (int)(qid % Q);
    const float* __restrict__ r = ref + (size_t)b * N * 3;
    const float qx = query[qid * 3 + 0], qy = query[qid * 3 + 1], qz = query[qid * 3 + 2];
    const float INF = __int_as_float(0x7f800000);
    const int per = (N + 63) / 64, k0 = lane * per, k1 = min(N, k0 + per);
    float last_d = -1.0f; int last_i = -1;
    for (int round = 0; round < K; ++round) {
        float lmin = INF; int li = 0;
        for (int k = k0; k < k1; ++k) {
            const float v = sqdist3(r[k * 3 + 0], r[k * 3 + 1], r[k * 3 + 2], qx, qy, qz);
            const bool after = (v > last_d) || (v == last_d && k > last_i);
            if (after && v < lmin) { lmin = v; li = k; }
        }
        const float m = wave_min_f32(lmin, INF);
        const int wl = first_lane(__ballot(lmin == m));
        const int widx = __builtin_amdgcn_readlane(li, wl);
        last_d = m; last_i = widx;
        if (lane == 0) {
            const size_t o = idx_kq ? ((size_t)b * K + round) * Q + q : (size_t)qid * K + round;
            idx_out[o] = (int64_t)widx;
            if (dist_out) dist_out[o] = sqrtf(m);
            if (nbr_out) {
                float* __restrict__ w = nbr_out + ((size_t)qid * K + round) * 3;
                w[0] = __fsub_rn(r[widx * 3 + 0], qx); w[1] = __fsub_rn(r[widx * 3 + 1], qy); w[2] = __fsub_rn(r[widx * 3 + 2], qz);
            }
        }
    }
}

template <int PPL>
static int launch_knn(const float* ref, const float* query, int B, int N, int Q, int K, int64_t* idx, int idx_kq,
                      float* nbr, float* dist, hipStream_t s) {
    const long long nq = (long long)B * Q;
    const int wpb = 4;
    static const bool two_level = [] { const char* e = getenv("ACT_KNN_TWO_LEVEL"); return !(e && e[0] == '0'); }();
    static const int tree = [] { const char* e = getenv("ACT_KNN_TREE"); return e ? atoi(e) : 1; }();   // 1: tournament-tree kernel (PPL >= 8); 0: two-level kernel (A/B)
    if constexpr (PPL >= 8) {
        if (tree) {
            hipLaunchKernelGGL(knn_tree_kernel<PPL>, dim3((unsigned)((nq + wpb - 1) / wpb)), dim3(wpb * 64), 0, s, ref, query,
                               B, N, Q, K, idx, idx_kq, nbr, dist);
            ACT_LAUNCH_CHECK();
            return 0;
        }
    }
    if constexpr (PPL >= 64) {
        if (two_level) {
            hipLaunchKernelGGL(knn_group2_kernel<PPL>, dim3((unsigned)((nq + wpb - 1) / wpb)), dim3(wpb * 64), 0, s, ref, query,
                               B, N, Q, K, idx, idx_kq, nbr, dist);
            ACT_LAUNCH_CHECK();
            return 0;
        }
    }
    // 8 .. 32 points per lane (N = 512 .. 2,048): groups of 4 -- retiring a winner rescans 4 distances and folds PPL / 4 group minima instead of
    // rescanning all PPL (the kernel is VALU-issue bound by exactly that rescan, executed by the whole wave for one active lane);
    // ACT_KNN_TWO_LEVEL_SMALL=0 selects the one-level kernel (A/B)
    static const bool two_level_small = [] { const char* e = getenv("ACT_KNN_TWO_LEVEL_SMALL"); return !(e && e[0] == '0'); }();
    if constexpr (PPL >= 8 && PPL <= 32) {
        if (two_level_small) {
            hipLaunchKernelGGL((knn_group2_kernel<PPL, 4>), dim3((unsigned)((nq + wpb - 1) / wpb)), dim3(wpb * 64), 0, s, ref, query,
                               B, N, Q, K, idx, idx_kq, nbr, dist);
            ACT_LAUNCH_CHECK();
            return 0;
        }
    }
    hipLaunchKernelGGL(knn_group_kernel<PPL>, dim3((unsigned)((nq + wpb - 1) / wpb)), dim3(wpb * 64), 0, s, ref, query,
                       B, N, Q, K, idx, idx_kq, nbr, dist);
    ACT_LAUNCH_CHECK();
    return 0;
}

extern "C" int act_knn_group_f32(const float* ref, const float* query, int B, int N, int Q, int K, int64_t* idx_out,
                                 int idx_kq, float* nbr_out, float* dist_out, act_stream_t stream) {
    if (B == 0 || Q == 0) return 0;                        // empty batch: nothing to do (empty tensors have NULL storage)
    if (!ref || !query || !idx_out) return ACT_E_NULLPTR;
    if (B < 0 || N <= 0 || Q < 0 || K <= 0 || K > N) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    // algorithmic bytes per cloud: 12N + 12Q + 8QK (+12QK neighbourhood) (+4QK dist)   [SURVEY 8d]
    ActProfScope ps(KID_KNN_GROUP, s, 0.0,
                    (double)B * (12.0 * N + 12.0 * Q + 8.0 * Q * K + (nbr_out ? 12.0 * Q * K : 0.0) + (dist_out ? 4.0 * Q * K : 0.0)));
    // the register path keeps the K winners one per lane (K <= 64); larger K (any N) and N > 8192 take the rescan kernel
#define KNN_CASE(P) if (K <= 64 && N <= 64 * P) return launch_knn<P>(ref, query, B, N, Q, K, idx_out, idx_kq, nbr_out, dist_out, s)
    KNN_CASE(1); KNN_CASE(2); KNN_CASE(4); KNN_CASE(8); KNN_CASE(16); KNN_CASE(32); KNN_CASE(64); KNN_CASE(128);
#undef KNN_CASE
    {
        const long long nq = (long long)B * Q;
        hipLaunchKernelGGL(knn_group_big_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, ref, query, B, N, Q, K,
                           idx_out, idx_kq, nbr_out, dist_out);
        ACT_LAUNCH_CHECK();
    }
    return 0;
}

// ============================================ gather ==============================================
__global__ void gather_points_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx, int C, int N, int S,
                                     float* __restrict__ out, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(i % S); const long long bc = i / S; const int b = (int)(bc / C);
        out[i] = feat[bc * N + idx[(size_t)b * S + s]];
    }
}
// deterministic scatter-add: one thread per (b,c,n) scans the S sampled indices (S is small: G)
__global__ void gather_points_bwd_kernel(const float* __restrict__ go, const int32_t* __restrict__ idx, int C, int N, int S,
                                         float* __restrict__ gf, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N); const long long bc = i / N; const int b = (int)(bc / C);
        const int32_t* __restrict__ id = idx + (size_t)b * S;
        float acc = 0.f;
        for (int s = 0; s < S; ++s) if (id[s] == n) acc += go[bc * S + s];
        gf[i] = acc;
    }
}
static inline unsigned grid_for(long long total, int block) {
    long long g = (total + block - 1) / block; if (g > 2048 * 4) g = 2048 * 4; if (g < 1) g = 1; return (unsigned)g;
}
extern "C" int act_gather_points_f32(const float* feat, const int32_t* idx, int B, int C, int N, int S, float* out, act_stream_t stream) {
    const long long total = (long long)B * C * S; if (total == 0) return 0;
    if (!feat || !idx || !out) return ACT_E_NULLPTR;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_GATHER, s, 0.0, 8.0 * total + 4.0 * B * S);
    hipLaunchKernelGGL(gather_points_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, feat, idx, C, N, S, out, total);
    ACT_LAUNCH_CHECK(); return 0;
}
extern "C" int act_gather_points_bwd_f32(const float* go, const int32_t* idx, int B, int C, int N, int S, float* gf, act_stream_t stream) {
    const long long total = (long long)B * C * N; if (total == 0) return 0;
    if (!go || !idx || !gf) return ACT_E_NULLPTR;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_GATHER_BWD, s, 0.0, 4.0 * total + 4.0 * B * C * S);
    hipLaunchKernelGGL(gather_points_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, go, idx, C, N, S, gf, total);
    ACT_LAUNCH_CHECK(); return 0;
}

// ======================================== augmentation ============================================
__global__ void scale_translate_kernel(float* __restrict__ pc, const float* __restrict__ scale, const float* __restrict__ shift,
                                       int N, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3); const int b = (int)(i / (3LL * N));
        pc[i] = __fadd_rn(__fmul_rn(pc[i], scale[b * 3 + c]), shift[b * 3 + c]);
    }
}
extern "C" int act_scale_translate_f32(float* pc, const float* scale, const float* shift, int B, int N, act_stream_t stream) {
    const long long total = (long long)B * N * 3; if (total == 0) return 0;
    if (!pc || !scale || !shift) return ACT_E_NULLPTR;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_AUGMENT, s, 0.0, 8.0 * total);
    hipLaunchKernelGGL(scale_translate_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, pc, scale, shift, N, total);
    ACT_LAUNCH_CHECK(); return 0;
}

// PointcloudRotate: p' = p @ R[b]  (out_j = sum_k p_k R[k][j]); one thread per point
__global__ void rotate_points_kernel(float* __restrict__ pc, const float* __restrict__ rot, int N, long long npts) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npts; i += (long long)gridDim.x * blockDim.x) {
        const float* r = rot + (i / N) * 9;
        const float x = pc[3 * i], y = pc[3 * i + 1], z = pc[3 * i + 2];
        pc[3 * i]     = x * r[0] + y * r[3] + z * r[6];
        pc[3 * i + 1] = x * r[1] + y * r[4] + z * r[7];
        pc[3 * i + 2] = x * r[2] + y * r[5] + z * r[8];
    }
}
extern "C" int act_rotate_points_f32(float* pc, const float* rot, int B, int N, act_stream_t stream) {
    const long long npts = (long long)B * N; if (npts == 0) return 0;
    if (!pc || !rot) return ACT_E_NULLPTR;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_AUGMENT, s, 0.0, 24.0 * npts);
    hipLaunchKernelGGL(rotate_points_kernel, dim3(grid_for(npts, 256)), dim3(256), 0, s, pc, rot, N, npts);
    ACT_LAUNCH_CHECK(); return 0;
}
