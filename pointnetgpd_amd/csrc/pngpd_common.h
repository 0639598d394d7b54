// Shared device helpers for libpngpd (gfx950 / CDNA4 only — no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/pngpd.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PNGPD_WAVE 64

// v_mfma_f32_32x32x2_f32: D(32x32) += A(32x2) * B(2x32), exact fp32 (fmaf chain).
//   lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
//   D register r of lane l is D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

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
