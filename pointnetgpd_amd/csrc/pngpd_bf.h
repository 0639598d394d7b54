// bf16 matrix-core operands for the opt-in reduced-precision modes (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//   NT = 1  plain bf16 operands (BASELINE configs[2])
//   NT = 3  "bf16x3": a*w ~ a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, each part bf16 (products exact in fp32)
// Operand fragment of the 32x32x16 instruction: lane (j = lane & 31, g = lane >> 5) supplies row/column j and the
// eight k indices 8g .. 8g+7 of the 16-wide k-step as one 16-byte register quad.
#pragma once
#include "pngpd_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ u16 bf16_bits(float x) { return __builtin_bit_cast(u16, (__bf16)x); }
__device__ __forceinline__ float bf16_val(u16 b) { return __uint_as_float(((unsigned)b) << 16); }

__device__ __forceinline__ void split2(float x, u16 &hi, u16 &lo) {
    hi = bf16_bits(x);
    lo = bf16_bits(x - bf16_val(hi));
}

__device__ __forceinline__ f32x16 mfma_bf(const f32x4 &a, const f32x4 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                   c, 0, 0, 0);
}

// Eight fp32 values (k order v[0..7]) -> the hi operand quad and, for NT == 3, the residual quad.
template <int NT>
__device__ __forceinline__ void bf_pack8(const float (&v)[8], f32x4 &hi, f32x4 &lo) {
    unsigned hw[4], lw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        u16 h0, h1, l0 = 0, l1 = 0;
        if (NT == 3) { split2(v[2 * p], h0, l0); split2(v[2 * p + 1], h1, l1); }
        else { h0 = bf16_bits(v[2 * p]); h1 = bf16_bits(v[2 * p + 1]); }
        hw[p] = (unsigned)h0 | ((unsigned)h1 << 16);
        lw[p] = (unsigned)l0 | ((unsigned)l1 << 16);
    }
    hi = f32x4{__uint_as_float(hw[0]), __uint_as_float(hw[1]), __uint_as_float(hw[2]), __uint_as_float(hw[3])};
    lo = f32x4{__uint_as_float(lw[0]), __uint_as_float(lw[1]), __uint_as_float(lw[2]), __uint_as_float(lw[3])};
}

template <int NT>
__device__ __forceinline__ void bf_pack8(const f32x4 &v0, const f32x4 &v1, f32x4 &hi, f32x4 &lo) {
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    bf_pack8<NT>(v, hi, lo);
}

// One k-step of the product in NT terms.
template <int NT>
__device__ __forceinline__ f32x16 bf_mma(const f32x4 &ah, const f32x4 &al, const f32x4 &bh, const f32x4 &bl, f32x16 c) {
    c = mfma_bf(ah, bh, c);
    if (NT == 3) { c = mfma_bf(ah, bl, c); c = mfma_bf(al, bh, c); }
    return c;
}

// B-operand fragments written by split_pack_bf16_kernel for a (C,K) row-major matrix W (out = in . W^T):
//   quad index ((cb*KS + ks)*2 + part)*64 + lane,   KS = K/16, part 0 = hi, 1 = lo
template <int NT>
__device__ __forceinline__ void bf_wfrag(const u16 *__restrict__ wx, int KS, int cb, int ks, int lane, f32x4 &wh, f32x4 &wl) {
    const f32x4 *p = (const f32x4 *)wx + ((size_t)(cb * KS + ks) * 2) * 64 + lane;
    wh = p[0];
    if (NT == 3) wl = p[64]; else wl = wh;
}
