// nt_pipe.hip -- main-loop experiments for the fp32 NT GEMM (C[M,N] = A[M,K] . B[N,K]^T, 128x128 tiles, v_mfma_f32_16x16x4_f32):
//   V0  the product kernel's loop (csrc/gemm16.hip sgemm_nt16_kernel): fragments read at the top of every K-tile, one __syncthreads
//   V1  software-pipelined: the fragments of tile t+1 are read from LDS while the MFMAs of tile t run (two register sets), global
//       prefetch distance 2, LDS-only wait + bare s_barrier so the global loads stay in flight across the barrier
// build: hipcc --offload-arch=gfx950 -O3 -o nt_pipe nt_pipe.hip ; run: ./nt_pipe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct P { const float* A; const float* B; float* C; int M, N, K, lda, ldb, ldc, tiles_m, tiles_n, group_m; };

__device__ __forceinline__ int xcd_remap(int wg, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, loc = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}
__device__ __forceinline__ void tile_coords(const P& p, int wg, int& tile_m, int& tile_n) {
    const int per_group = p.group_m * p.tiles_n;
    const int group = wg / per_group, first_m = group * p.group_m;
    const int gsz = min(p.tiles_m - first_m, p.group_m);
    const int in_group = wg - group * per_group;
    tile_m = first_m + in_group % gsz;
    tile_n = in_group / gsz;
}

struct Frag { float4 a[4], b[4]; };

__device__ __forceinline__ void mfma_tile(f32x4 (&acc)[4][4], const Frag& f) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i].x, f.b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i].y, f.b[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i].z, f.b[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i].w, f.b[j].w, acc[i][j], 0, 0, 0);
}

template <int PIN>
__device__ __forceinline__ void lds_barrier() {          // LDS traffic of this wave done, then the workgroup barrier; global loads stay in flight
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);   // keep the phase's MFMAs in the phase (hipcc otherwise moves them across)
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0), vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0); else asm volatile("" ::: "memory");
}

template <int V, int PIN = 0, int ORD = 0>
__global__ __launch_bounds__(256, 3) void nt_kernel(const P p) {
    constexpr int BM = 128, BN = 128, BK = 16;
    __shared__ __attribute__((aligned(16))) float As[2][BM * BK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int wg = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    int tile_m, tile_n;
    tile_coords(p, wg, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int ntiles = p.K / BK;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int srow = tid >> 2, sch = tid & 3;
    const float* ga = p.A + (size_t)(m0 + srow) * p.lda + sch * 4;
    const float* gb = p.B + (size_t)(n0 + srow) * p.ldb + sch * 4;
    const int s_off = srow * 16 + 4 * (sch ^ ((4 - ((srow >> 2) & 3)) & 3));
    const size_t stride_a = (size_t)64 * p.lda, stride_b = (size_t)64 * p.ldb;
    float4 ra0, ra1, rb0, rb1;
    auto load_g = [&](int t) {
        ra0 = *reinterpret_cast<const float4*>(ga + t * BK);
        ra1 = *reinterpret_cast<const float4*>(ga + stride_a + t * BK);
        rb0 = *reinterpret_cast<const float4*>(gb + t * BK);
        rb1 = *reinterpret_cast<const float4*>(gb + stride_b + t * BK);
    };
    auto store_lds = [&](int buf) {
        *reinterpret_cast<float4*>(&As[buf][s_off]) = ra0;
        *reinterpret_cast<float4*>(&As[buf][s_off + 1024]) = ra1;
        *reinterpret_cast<float4*>(&Bs[buf][s_off]) = rb0;
        *reinterpret_cast<float4*>(&Bs[buf][s_off + 1024]) = rb1;
    };
    const int kl = lane >> 4, ml = lane & 15;
    const int hsw = (4 - ((ml >> 2) & 3)) & 3;
    const int a_off = (wm * 64 + ml) * 16 + 4 * (kl ^ hsw);
    const int b_off = (wn * 64 + ml) * 16 + 4 * (kl ^ hsw);
    auto read_frags = [&](Frag& f, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f.a[i] = *reinterpret_cast<const float4*>(&As[buf][a_off + i * 256]);
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j] = *reinterpret_cast<const float4*>(&Bs[buf][b_off + j * 256]);
    };

    if constexpr (V >= 10) {          // ablations of V0: 10 = MFMAs only, 11 = + fragment reads, 12 = + barrier, 13 = + LDS writes (no global loads in the loop)
        load_g(0); store_lds(0); store_lds(1); __syncthreads();
        Frag f; read_frags(f, 0);
        for (int t = 0; t + 1 < ntiles; ++t) {
            if constexpr (V >= 11) read_frags(f, t & 1);
            mfma_tile(acc, f);
            if constexpr (V >= 13) { ra0.x += 1.f; store_lds((t & 1) ^ 1); }
            if constexpr (V >= 12) __syncthreads(); else asm volatile("" ::: "memory");
        }
        mfma_tile(acc, f);
    } else if constexpr (V == 0) {
        load_g(0); store_lds(0); __syncthreads();
        for (int t = 0; t + 1 < ntiles; ++t) {
            load_g(t + 1);
            Frag f; read_frags(f, t & 1); mfma_tile(acc, f);
            store_lds((t & 1) ^ 1);
            __syncthreads();
        }
        Frag f; read_frags(f, (ntiles - 1) & 1); mfma_tile(acc, f);
    } else if constexpr (V == 1) {
        // branch-free steady state (ntiles even): loads past the end are clamped to the last tile and land in an LDS buffer nobody reads
        auto order1 = [&]() {
            if constexpr (!ORD) return;             // fragment reads first, then the LDS writes of the tile loaded one phase ago, then its successor's loads
#pragma unroll
            for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 40, 0);
        };
        Frag F0, F1;
        const int last = ntiles - 1;
        load_g(0); store_lds(0);
        load_g(min(1, last));
        lds_barrier<PIN>();
        read_frags(F0, 0);
        store_lds(1); load_g(min(2, last));
        lds_barrier<PIN>();
        for (int t = 0; t < ntiles; t += 2) {
            read_frags(F1, 1);
            mfma_tile(acc, F0);
            store_lds(0); load_g(min(t + 3, last));
            order1();
            lds_barrier<PIN>();
            read_frags(F0, 0);
            mfma_tile(acc, F1);
            store_lds(1); load_g(min(t + 4, last));
            order1();
            lds_barrier<PIN>();
        }
    }
    if constexpr (V == 2) {
        // V2: V1 with LDS-DMA staging (global_load_lds_dwordx4: wave-uniform LDS base + lane * 16, so the XOR swizzle moves to the per-lane
        // SOURCE chunk): no staging registers, no ds_write pass; tile t+2 is in flight to the buffer tile t left while tile t is multiplied
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int swz = sch ^ ((4 - ((srow >> 2) & 3)) & 3);
        const float* sa = p.A + (size_t)(m0 + srow) * p.lda + swz * 4;
        const float* sb = p.B + (size_t)(n0 + srow) * p.ldb + swz * 4;
        const int wbase = __builtin_amdgcn_readfirstlane(wave) * 256;
        auto glds = [&](int t, int buf) {
            __builtin_amdgcn_global_load_lds((gptr_t)(sa + t * BK), (lptr_t)&As[buf][wbase], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(sa + stride_a + t * BK), (lptr_t)&As[buf][wbase + 1024], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(sb + t * BK), (lptr_t)&Bs[buf][wbase], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(sb + stride_b + t * BK), (lptr_t)&Bs[buf][wbase + 1024], 16, 0, 0);
        };
        auto full_barrier = [&]() {
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0070);              // vmcnt(0) lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            if constexpr (PIN) __builtin_amdgcn_sched_barrier(0); else asm volatile("" ::: "memory");
        };
        auto phase_order = [&]() {
            if constexpr (!ORD) return;                           // LDS reads and the LDS-DMA issue first, one MFMA between each, then the rest
#pragma unroll
            for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 48, 0);
        };
        Frag F0, F1;
        const int last = ntiles - 1;
        glds(0, 0);
        full_barrier();
        read_frags(F0, 0);
        glds(min(1, last), 1);
        full_barrier();
        for (int t = 0; t < ntiles; t += 2) {
            read_frags(F1, 1);
            glds(min(t + 2, last), 0);
            mfma_tile(acc, F0);
            phase_order();
            full_barrier();
            read_frags(F0, 0);
            glds(min(t + 3, last), 1);
            mfma_tile(acc, F1);
            phase_order();
            full_barrier();
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + ml;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + kl * 4 + r;
                p.C[(size_t)row * p.ldc + col] = acc[i][j][r];
            }
        }
}

template <int V, int PIN = 0, int ORD = 0>
static double run(const P& p, int reps) {
    dim3 grid(p.tiles_m * p.tiles_n);
    hipLaunchKernelGGL((nt_kernel<V, PIN, ORD>), grid, dim3(256), 0, 0, p);
    HIPCHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    double best = 1e9;
    for (int r = 0; r < 3; ++r) {
        HIPCHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((nt_kernel<V, PIN, ORD>), grid, dim3(256), 0, 0, p);
        HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
        float ms; HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / reps < best) best = ms / reps;
    }
    return best;
}

// ---- larger workgroup tiles (fewer L2->LDS bytes and fewer LDS fragment reads per MFMA: the chip clocks to its power budget) -----------
// BM x BN tile, WM x WN waves, every wave owns (BM/WM) x (BN/WN); same LDS image / swizzle / loop as V0.
template <int BM, int BN, int WM, int WN, int OCC>
__global__ __launch_bounds__(WM * WN * 64, OCC) void nt_big(const P p) {
    constexpr int BK = 16, NT = WM * WN * 64, TM = BM / WM / 16, TN = BN / WN / 16, RP = NT / 4, NA = BM / RP, NB = BN / RP;
    __shared__ __attribute__((aligned(16))) float As[2][BM * BK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int wg = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    int tile_m, tile_n;
    tile_coords(p, wg, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int ntiles = p.K / BK;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int srow = tid >> 2, sch = tid & 3;
    const float* ga = p.A + (size_t)(m0 + srow) * p.lda + sch * 4;
    const float* gb = p.B + (size_t)(n0 + srow) * p.ldb + sch * 4;
    const int s_off = srow * 16 + 4 * (sch ^ ((4 - ((srow >> 2) & 3)) & 3));
    float4 ra[NA], rb[NB];
    const int kl = lane >> 4, ml = lane & 15;
    const int hsw = (4 - ((ml >> 2) & 3)) & 3;
    const int a_off = (wm * (BM / WM) + ml) * 16 + 4 * (kl ^ hsw);
    const int b_off = (wn * (BN / WN) + ml) * 16 + 4 * (kl ^ hsw);
#define LOAD_G(t) { _Pragma("unroll") for (int q = 0; q < NA; ++q) ra[q] = *reinterpret_cast<const float4*>(ga + (size_t)q * RP * p.lda + (t) * BK); \
                    _Pragma("unroll") for (int q = 0; q < NB; ++q) rb[q] = *reinterpret_cast<const float4*>(gb + (size_t)q * RP * p.ldb + (t) * BK); }
#define STORE_L(buf) { _Pragma("unroll") for (int q = 0; q < NA; ++q) *reinterpret_cast<float4*>(&As[buf][s_off + q * RP * 16]) = ra[q]; \
                       _Pragma("unroll") for (int q = 0; q < NB; ++q) *reinterpret_cast<float4*>(&Bs[buf][s_off + q * RP * 16]) = rb[q]; }
#define COMPUTE(buf) { float4 af[TM], bf[TN]; \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(&As[buf][a_off + i * 256]); \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bs[buf][b_off + j * 256]); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0); }
    LOAD_G(0); STORE_L(0); __syncthreads();
    for (int t = 0; t + 1 < ntiles; t += 2) {           // ntiles even
        LOAD_G(t + 1); COMPUTE(0); STORE_L(1); __syncthreads();
        if (t + 2 < ntiles) { LOAD_G(t + 2); }
        COMPUTE(1);
        if (t + 2 < ntiles) { STORE_L(0); }
        __syncthreads();
    }
#undef LOAD_G
#undef STORE_L
#undef COMPUTE
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 16 + ml;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * (BM / WM) + i * 16 + kl * 4 + r;
                p.C[(size_t)row * p.ldc + col] = acc[i][j][r];
            }
        }
}

template <int BM, int BN, int WM, int WN, int OCC>
static double run_big(P p, int reps) {
    p.tiles_m = p.M / BM; p.tiles_n = p.N / BN;
    dim3 grid(p.tiles_m * p.tiles_n);
    hipLaunchKernelGGL((nt_big<BM, BN, WM, WN, OCC>), grid, dim3(WM * WN * 64), 0, 0, p);
    HIPCHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    double best = 1e9;
    for (int r = 0; r < 3; ++r) {
        HIPCHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((nt_big<BM, BN, WM, WN, OCC>), grid, dim3(WM * WN * 64), 0, 0, p);
        HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
        float ms; HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / reps < best) best = ms / reps;
    }
    return best;
}

// ---- persistent V0: 768 workgroups (3 per CU), each walks tiles blockIdx.x, +gridDim.x, ... (no workgroup relaunch between tiles) -------
template <int MODE>
__global__ __launch_bounds__(256, 3) void nt_persist(const P p) {
    constexpr int BM = 128, BN = 128, BK = 16;
    __shared__ __attribute__((aligned(16))) float As[2][BM * BK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int total = p.tiles_m * p.tiles_n;
    const int ntiles = p.K / BK;
    const int srow = tid >> 2, sch = tid & 3;
    const int s_off = srow * 16 + 4 * (sch ^ ((4 - ((srow >> 2) & 3)) & 3));
    const size_t stride_a = (size_t)64 * p.lda, stride_b = (size_t)64 * p.ldb;
    const int kl = lane >> 4, ml = lane & 15;
    const int hsw = (4 - ((ml >> 2) & 3)) & 3;
    const int a_off = (wm * 64 + ml) * 16 + 4 * (kl ^ hsw);
    const int b_off = (wn * 64 + ml) * 16 + 4 * (kl ^ hsw);
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        const int wg = xcd_remap(w, total);
        int tile_m, tile_n;
        tile_coords(p, wg, tile_m, tile_n);
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* ga = p.A + (size_t)(m0 + srow) * p.lda + sch * 4;
        const float* gb = p.B + (size_t)(n0 + srow) * p.ldb + sch * 4;
        float4 ra0, ra1, rb0, rb1;
        auto load_g = [&](int t) {
            ra0 = *reinterpret_cast<const float4*>(ga + t * BK);
            ra1 = *reinterpret_cast<const float4*>(ga + stride_a + t * BK);
            rb0 = *reinterpret_cast<const float4*>(gb + t * BK);
            rb1 = *reinterpret_cast<const float4*>(gb + stride_b + t * BK);
        };
        auto store_lds = [&](int buf) {
            *reinterpret_cast<float4*>(&As[buf][s_off]) = ra0;
            *reinterpret_cast<float4*>(&As[buf][s_off + 1024]) = ra1;
            *reinterpret_cast<float4*>(&Bs[buf][s_off]) = rb0;
            *reinterpret_cast<float4*>(&Bs[buf][s_off + 1024]) = rb1;
        };
        auto read_frags = [&](Frag& f, int buf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) f.a[i] = *reinterpret_cast<const float4*>(&As[buf][a_off + i * 256]);
#pragma unroll
            for (int j = 0; j < 4; ++j) f.b[j] = *reinterpret_cast<const float4*>(&Bs[buf][b_off + j * 256]);
        };
        load_g(0); store_lds(0); __syncthreads();
        for (int t = 0; t + 1 < ntiles; ++t) {
            load_g(t + 1);
            Frag f; read_frags(f, t & 1); mfma_tile(acc, f);
            store_lds((t & 1) ^ 1);
            __syncthreads();
        }
        { Frag f; read_frags(f, (ntiles - 1) & 1); mfma_tile(acc, f); }
        __syncthreads();                                 // the next tile's first store_lds(0) must not overtake a slow wave's last reads
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = n0 + wn * 64 + j * 16 + ml;
#pragma unroll
                    for (int r = 0; r < 4; ++r) p.C[(size_t)(m0 + wm * 64 + i * 16 + kl * 4 + r) * p.ldc + col] = acc[i][j][r];
                }
        }
    }
}
template <int MODE>
static double run_persist(const P& p, int reps, int nwg) {
    dim3 grid(nwg);
    hipLaunchKernelGGL((nt_persist<MODE>), grid, dim3(256), 0, 0, p);
    HIPCHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    double best = 1e9;
    for (int r = 0; r < 3; ++r) {
        HIPCHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((nt_persist<MODE>), grid, dim3(256), 0, 0, p);
        HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
        float ms; HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / reps < best) best = ms / reps;
    }
    return best;
}

int main() {
    const int shapes[][3] = {{16384, 3072, 768}, {16384, 768, 3072}, {8192, 3072, 768}, {8192, 768, 3072}, {16384, 2304, 768}, {8192, 768, 768}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        float *dA, *dB, *dC0, *dC1;
        HIPCHECK(hipMalloc(&dA, (size_t)M * K * 4)); HIPCHECK(hipMalloc(&dB, (size_t)N * K * 4));
        HIPCHECK(hipMalloc(&dC0, (size_t)M * N * 4)); HIPCHECK(hipMalloc(&dC1, (size_t)M * N * 4));
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
        unsigned s = 12345u;
        for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
        for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
        HIPCHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        P p{dA, dB, dC0, M, N, K, K, K, N, M / 128, N / 128, 8};
        const double flops = 2.0 * M * N * (double)K;
        const double t0 = run<0>(p, 10);
        std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
        HIPCHECK(hipMemcpy(c0.data(), dC0, c0.size() * 4, hipMemcpyDeviceToHost));
        p.C = dC1;
        printf("%6d x %5d x %5d  V0 %.3f ms %.1f TF\n", M, N, K, t0, flops / t0 / 1e9);
        auto report = [&](const char* name, double t) {
            HIPCHECK(hipMemcpy(c1.data(), dC1, c1.size() * 4, hipMemcpyDeviceToHost));
            double md = 0; for (size_t i = 0; i < c0.size(); ++i) md = fmax(md, fabs((double)c0[i] - c1[i]));
            printf("      %-24s %.3f ms %6.1f TF  max|d| %g\n", name, t, flops / t / 1e9, md);
            HIPCHECK(hipMemset(dC1, 0, (size_t)M * N * 4));
        };
        report("abl: mfma only", run<10>(p, 10));
        report("abl: + frag reads", run<11>(p, 10));
        report("abl: + barrier", run<12>(p, 10));
        report("abl: + lds writes", run<13>(p, 10));
        report("V0 again", run<0>(p, 10));
        report("V0 persistent 768", run_persist<0>(p, 10, 768));
        report("V0 persistent 512", run_persist<0>(p, 10, 512));
        report("V1 regs", run<1, 0, 0>(p, 10));
        report("V1 regs pin", run<1, 1, 0>(p, 10));
        report("V1 regs pin+order", run<1, 1, 1>(p, 10));
        report("V2 lds-dma", run<2, 0, 0>(p, 10));
        report("V2 lds-dma pin", run<2, 1, 0>(p, 10));
        report("V2 lds-dma pin+order", run<2, 1, 1>(p, 10));
        HIPCHECK(hipFree(dA)); HIPCHECK(hipFree(dB)); HIPCHECK(hipFree(dC0)); HIPCHECK(hipFree(dC1));
    }
    return 0;
}
