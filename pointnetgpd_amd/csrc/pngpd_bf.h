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

// Two fp32 values -> one dword of two bf16 (element 0 in the low half): ONE v_cvt_pk_bf16_f32.  (Round 6: the scalar
// form — (__bf16)x per value, halves merged with shifts and ors — compiled to one v_cvt_pk_bf16_f32 PER VALUE plus a
// v_lshlrev / v_or_b32_sdwa pair per dword: 3.5 VALU instructions per packed pair, 272 + 120 + 152 of the 1,270 VALU
// instructions of a pass-D tile in plain-bf16 mode.  Same instruction, same round-to-nearest-even: results unchanged.)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf_pk2(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// hi dword as above and the dword of the two residuals x - float(hi) (the bf16x3 split of a pair: 5 instructions)
__device__ __forceinline__ void bf_split_pk2(float a, float b, unsigned &hi, unsigned &lo) {
    hi = bf_pk2(a, b);
    const f32x2 r = f32x2{a, b} - f32x2{__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
    lo = bf_pk2(r[0], r[1]);
}

// Eight fp32 values (k order v[0..7]) -> the hi operand quad and, for NT == 3, the residual quad.
template <int NT>
__device__ __forceinline__ void bf_pack8(const float (&v)[8], f32x4 &hi, f32x4 &lo) {
    unsigned hw[4], lw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (NT == 3) bf_split_pk2(v[2 * p], v[2 * p + 1], hw[p], lw[p]);
        else { hw[p] = bf_pk2(v[2 * p], v[2 * p + 1]); lw[p] = 0u; }
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

// ---- bf16 storage of the lane-major z2 / g2 tiles (plain-bf16 training mode, NT == 1) --------------------------------
// fp32 tiles: [(b*T + tile)][8][256] float4, value v = 4q + e of thread tid.  bf16 tiles hold the same 32 values per
// thread as [(b*T + tile)][4][256] uint4: value v = 8i + e8 is half-word e8 of quad i — half the bytes, four 16-byte
// accesses per lane instead of eight.
__device__ __forceinline__ uint4 bf_tile_pack(const float (&v)[8]) {
    uint4 o;
    o.x = bf_pk2(v[0], v[1]);
    o.y = bf_pk2(v[2], v[3]);
    o.z = bf_pk2(v[4], v[5]);
    o.w = bf_pk2(v[6], v[7]);
    return o;
}
__device__ __forceinline__ void bf_tile_unpack(const uint4 &q, float (&v)[8]) {
    v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
    v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
    v[4] = __uint_as_float(q.z << 16); v[5] = __uint_as_float(q.z & 0xffff0000u);
    v[6] = __uint_as_float(q.w << 16); v[7] = __uint_as_float(q.w & 0xffff0000u);
}
// the thread's 32 tile values <-> two MFMA register sets (a0 = values 0..15, a1 = values 16..31)
__device__ __forceinline__ void bf_tile_load(const uint4 *__restrict__ t, f32x16 &a0, f32x16 &a1) {   // t: + i*256 per quad
    uint4 q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = t[(size_t)i * 256];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v[8];
        bf_tile_unpack(q[i], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { if (i < 2) a0[8 * i + e] = v[e]; else a1[8 * (i - 2) + e] = v[e]; }
    }
}
__device__ __forceinline__ void bf_tile_store(uint4 *__restrict__ t, const f32x16 &a0, const f32x16 &a1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (i < 2) ? a0[8 * i + e] : a1[8 * (i - 2) + e];
        t[(size_t)i * 256] = bf_tile_pack(v);
    }
}
