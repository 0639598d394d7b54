// Shared device helpers for libpngpd (gfx950 / CDNA4 only — no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/pngpd.h"

// AddressSanitizer builds (`make asan`): the kernels that run 1024-thread workgroups in the product build get 256-thread
// workgroups — an instrumented 1024-thread kernel is held to 128 VGPRs and spills ~1,700 registers per lane
// (profiles/r03_asan.txt); same algorithm, fewer row lanes (the summation order differs from the product build's).
#if defined(__SANITIZE_ADDRESS__)
#define PNGPD_ASAN 1
#elif defined(__has_feature)
#if __has_feature(address_sanitizer)
#define PNGPD_ASAN 1
#endif
#endif
#ifndef PNGPD_ASAN
#define PNGPD_ASAN 0
#endif
#define PNGPD_RED_RL (PNGPD_ASAN ? 8 : 32)     // row lanes of the 32-column reduction kernels
#define PNGPD_BN3_RL (PNGPD_ASAN ? 16 : 64)    // row lanes of bn3_bwd_prep_kernel (16 channels per workgroup)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define PNGPD_WAVE 64

// v_mfma_f32_32x32x2_f32: D(32x32) += A(32x2) * B(2x32), exact fp32 (fmaf chain).
//   lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
//   D register r of lane l is D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// The two 32-lane halves of a wave exchange a register: lo = the value held by lane (l & 31), hi = the value held by
// lane 32 + (l & 31), in every lane — one v_permlane32_swap_b32 (a VALU instruction of gfx950) where __shfl_xor(x, 32)
// is a ds_bpermute_b32, i.e. an LDS round trip queued behind the workgroup's A-fragment reads.
__device__ __forceinline__ void half_pair(float x, float &lo, float &hi) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
}
__device__ __forceinline__ void half_pair(int x, int &lo, int &hi) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    lo = (int)r[0]; hi = (int)r[1];
}
__device__ __forceinline__ float half_sum(float x) { float lo, hi; half_pair(x, lo, hi); return lo + hi; }

// ---- the per-lane epilogue of a pair of 32x32 accumulator blocks (training pass C) -----------------------------
// On gfx950 the fp32 matrix instruction does NOT overlap VALU work on its SIMD, from either resident wave
// (tools/probes/mfma_valu_overlap.hip: a v_mfma_f32_32x32x2_f32 costs 64 cycles, every plain VALU instruction ~4.7
// more, every v_cmp ~8, and the times ADD) — so an epilogue is paid in matrix-pipe time, instruction by instruction.
// lane_max_moments() gets the exact first maximum of the lane's 32 values (rows mfma_row(r, lane) of a0 and
// 32 + mfma_row(r, lane) of a1, ascending) and their sum / sum of squares in ~120 VALU instructions and no compares,
// where the compare-and-select chain took ~200 including 64 v_cmp:
//   m    = v_max3 tree                                                (17)
//   su,qu= packed adds / packed FMAs on register pairs                (34)
//   row  = min over r of ((bits(m - v_r) & ~63) | code_r): m - v_r is +0 exactly for the maxima, so the smallest key
//          is the smallest row code among them (packed subtract 16, v_and_or 32, v_min3_u32 tree 16).
// Non-finite maxima (m - v = NaN) find no key below 64: the row is then arbitrary (callers clamp it); such a step has
// already diverged.  Returns the row WITHOUT the lane half's +4 (mfma_row's 4 * (lane >> 5)).
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ unsigned min3u(unsigned a, unsigned b, unsigned c) { return min(min(a, b), c); }

__device__ __forceinline__ void lane_max_moments(const f32x16 &a0, const f32x16 &a1, float &m, int &row,
                                                 float &su, float &qu) {
    float v[32];
#pragma unroll
    for (int r = 0; r < 16; ++r) { v[r] = a0[r]; v[16 + r] = a1[r]; }
    float t[11];
#pragma unroll
    for (int i = 0; i < 10; ++i) t[i] = max3f(v[3 * i], v[3 * i + 1], v[3 * i + 2]);
    t[10] = __builtin_fmaxf(v[30], v[31]);
    m = max3f(max3f(t[0], t[1], t[2]), max3f(t[3], t[4], t[5]),
              max3f(max3f(t[6], t[7], t[8]), t[9], t[10]));
    f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
    const f32x2 m2 = {m, m};
    unsigned k[32];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const f32x2 p = {v[2 * i], v[2 * i + 1]};
        s2 += p;
        q2 = __builtin_elementwise_fma(p, p, q2);
        const f32x2 d = m2 - p;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 2 * i + e;
            const unsigned code = (unsigned)(((r & 15) & 3) + 8 * ((r & 15) >> 2) + 32 * (r >> 4));
            k[r] = (__float_as_uint(d[e]) & 0xffffffc0u) | code;
        }
    }
    unsigned u[11];
#pragma unroll
    for (int i = 0; i < 10; ++i) u[i] = min3u(k[3 * i], k[3 * i + 1], k[3 * i + 2]);
    u[10] = min(k[30], k[31]);
    const unsigned kk = min3u(min3u(u[0], u[1], u[2]), min3u(u[3], u[4], u[5]),
                              min3u(min3u(u[6], u[7], u[8]), u[9], u[10]));
    row = (int)(kk & 63u);
    su = s2[0] + s2[1];
    qu = q2[0] + q2[1];
}


// Dynamic LDS above 48 KB needs a per-kernel opt-in.  Idempotent and checked on every call: no "already set"
// flag to race on, and a failure is reported instead of surfacing later as a launch error.
static inline int pngpd_allow_lds(const void *fn, size_t bytes) {
    if (bytes <= 48 * 1024) return PNGPD_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return e == hipSuccess ? PNGPD_OK : (PNGPD_ERR_HIP + (int)e);
}

// Workgroups per cloud for a launch that aims at `target` workgroups: ceil(target / B) clamped to [1, T].
static inline int pngpd_splits_for(int B, int T, int target) {
    int S = (target + B - 1) / B;
    if (S < 1) S = 1;
    if (S > T) S = T;
    return S;
}

// fp64 operations that must round exactly like numpy's separate multiply / add.  HIP's __dmul_rn / __dadd_rn are
// plain operators defined in a header under the default -ffp-contract=fast-honor-pragmas: once inlined, hipcc fuses
// them into v_fma_f64 (measured: the crop kernel held 96 v_fmac_f64).  These helpers carry contract(off), so neither
// half of a*b + c may be fused.
__device__ __forceinline__ double pn_dmul(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ double pn_dadd(double a, double b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ double pn_dsub(double a, double b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ double pn_ddiv(double a, double b) {
#pragma clang fp contract(off)
    return a / b;
}

static inline int pngpd_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PNGPD_OK : (PNGPD_ERR_HIP + (int)e);
}

// Per-phase cycle accounting for kernel experiments (variant builds only: -DPNGPD_TIMING -fgpu-rdc is NOT needed: every
// translation unit accumulates into its own copy read back by pngpd_tm_read / pngpd_tm_read_x3; tools/phase_times.py;
// never in the product library).
#ifdef PNGPD_TIMING
#define TM_DECL unsigned long long tm_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tm_t = __builtin_amdgcn_s_memtime();
#define TM(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tm_[i] += n_ - tm_t; tm_t = n_; }
#define TM_END_TO(arr) if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&arr[i_], tm_[i_]); atomicAdd(&arr[15], 1ull); }
#define TM_END TM_END_TO(pngpd_tm)
#else
#define TM_DECL
#define TM(i)
#define TM_END
#define TM_END_TO(arr)
#endif

// pass A of the training trunk: one cloud's input moments in fp64, mom[b] = {sx,sy,sz, sxx,sxy,sxz, syy,syz,szz}; one
// 256-thread workgroup per cloud.  A device function because two launches carry it: cloud_moments_kernel
// (pngpd_cloud_moments) and, in the fused forward, the tail workgroups of the weight re-layout launch (train_pack_kernel).
__device__ __forceinline__ void cloud_moments_body(const float *__restrict__ x, int N, int b, double *__restrict__ mom) {
    __shared__ double red[4][9];
    const int tid = threadIdx.x;
    const float *xb = x + (size_t)b * 3 * N;
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int n = tid; n < N; n += 256) {
        double x0 = xb[n], x1 = xb[N + n], x2 = xb[2 * N + n];
        a[0] += x0; a[1] += x1; a[2] += x2;
        a[3] += x0 * x0; a[4] += x0 * x1; a[5] += x0 * x2;
        a[6] += x1 * x1; a[7] += x1 * x2; a[8] += x2 * x2;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) a[i] += __shfl_xor(a[i], m);
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) red[tid >> 6][i] = a[i];
    }
    __syncthreads();
    if (tid < 9) mom[(size_t)b * 9 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}
