// common.h -- shared device helpers + launch/profiling plumbing for libact_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/act_hip.h"

#define ACT_WAVE 64

// ---- live profiler (prof.hip) -------------------------------------------------------------------
enum ActKernelId {
    KID_FPS = 0, KID_KNN_GROUP, KID_GATHER, KID_GATHER_BWD, KID_AUGMENT, KID_CHAMFER_FWD, KID_CHAMFER_BWD,
    KID_GEMM_NT, KID_GEMM_NN, KID_GEMM_TN, KID_LAYERNORM_FWD, KID_LAYERNORM_BWD, KID_ATTN_FWD, KID_ATTN_BWD,
    KID_COLSUM, KID_GELU_BWD, KID_COSINE_FWD, KID_COSINE_BWD, KID_BN_STATS, KID_BN_APPLY, KID_BN_BWD,
    KID_MAXPOOL, KID_MAXPOOL_BWD, KID_GN_LRELU_MAX, KID_GRAPH_FEATURE, KID_GUMBEL_ARGMAX, KID_ROW_GATHER,
    KID_ROW_SCATTER, KID_ADAMW, KID_ELTWISE, KID_GEMM_BF16X3, KID_COUNT
};

void act_prof_begin(int kid, hipStream_t s, double flops, double bytes);
void act_prof_end(int kid, hipStream_t s);
extern int g_act_prof_on;
// fraction of non-zero int32 flags (group-liveness lists), read back SYNCHRONOUSLY: only while the instrumented pass of bench.py is on -- never on the hot
// path -- so that the byte model of a kernel that skips dead rows counts the rows it touches; 1.0 when profiling is off or flags == null
double act_prof_live_fraction(const int32_t* flags, int n, hipStream_t s);

struct ActProfScope {
    int kid; hipStream_t s; bool on;
    ActProfScope(int k, hipStream_t st, double flops, double bytes) : kid(k), s(st), on(g_act_prof_on != 0) {
        if (on) act_prof_begin(kid, s, flops, bytes);
    }
    ~ActProfScope() { if (on) act_prof_end(kid, s); }
};

#define ACT_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

// ---- exact fp32 helpers (one rounding per op, never contracted to FMA) -----------------------------
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ---- DPP wave reductions (64 lanes; result broadcast through an SGPR) -------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov_f(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// max over the wave; every value must be > ident
__device__ __forceinline__ float wave_max_f32(float v, float ident) {
    v = fmaxf(v, dpp_mov_f<0x111, 0xf>(ident, v));   // row_shr:1
    v = fmaxf(v, dpp_mov_f<0x112, 0xf>(ident, v));   // row_shr:2
    v = fmaxf(v, dpp_mov_f<0x114, 0xf>(ident, v));   // row_shr:4
    v = fmaxf(v, dpp_mov_f<0x118, 0xf>(ident, v));   // row_shr:8
    v = fmaxf(v, dpp_mov_f<0x142, 0xa>(ident, v));   // row_bcast:15 -> rows 1,3
    v = fmaxf(v, dpp_mov_f<0x143, 0xc>(ident, v));   // row_bcast:31 -> rows 2,3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min_f32(float v, float ident) {
    v = fminf(v, dpp_mov_f<0x111, 0xf>(ident, v));
    v = fminf(v, dpp_mov_f<0x112, 0xf>(ident, v));
    v = fminf(v, dpp_mov_f<0x114, 0xf>(ident, v));
    v = fminf(v, dpp_mov_f<0x118, 0xf>(ident, v));
    v = fminf(v, dpp_mov_f<0x142, 0xa>(ident, v));
    v = fminf(v, dpp_mov_f<0x143, 0xc>(ident, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_mov_f<0x111, 0xf>(0.f, v);
    v += dpp_mov_f<0x112, 0xf>(0.f, v);
    v += dpp_mov_f<0x114, 0xf>(0.f, v);
    v += dpp_mov_f<0x118, 0xf>(0.f, v);
    v += dpp_mov_f<0x142, 0xa>(0.f, v);
    v += dpp_mov_f<0x143, 0xc>(0.f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// integer max over the wave (no float canonicalisation ops in the dependent chain); ident must be <= every value
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov_i(int identity, int v) {
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_max_i32(int v, int ident) {
    v = max(v, dpp_mov_i<0x111, 0xf>(ident, v));
    v = max(v, dpp_mov_i<0x112, 0xf>(ident, v));
    v = max(v, dpp_mov_i<0x114, 0xf>(ident, v));
    v = max(v, dpp_mov_i<0x118, 0xf>(ident, v));
    v = max(v, dpp_mov_i<0x142, 0xa>(ident, v));
    v = max(v, dpp_mov_i<0x143, 0xc>(ident, v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int first_lane(unsigned long long ballot) { return __ffsll((long long)ballot) - 1; }

// ---------------------------------------------------------------------------------------------- Philox4x32-10 (counter RNG)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
