// Tile-level building blocks shared by the inference and training trunk kernels.
// A workgroup = 256 threads = 4 waves; a tile = 64 points of one cloud.
//   xs  [3][64]   staged (optionally T^T-transformed) coordinates
//   h1  [64][68]  layer-1 activations  (row = point, 64 channels + 4 pad floats)
//   h2  [64][132] layer-2 activations  (row = point, 128 channels + 4 pad floats)
// The +4-float row pad makes the per-lane float4 A-fragment reads (ds_read_b128: lane i reads
// row i at a common column offset) conflict-free: row stride 68 / 132 dwords == 4 (mod 64).
#pragma once
#include "pngpd_common.h"
#include "pngpd_bf.h"

#define TP 64
#define H1S 68
#define H2S 132

struct Lane {
    int tid, lane, wave, j, h;
    __device__ __forceinline__ Lane() {
        tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        j = lane & 31;
        h = lane >> 5;
    }
};

// Load the tile's 64 points (tail: replicate point N-1), optionally apply x' = x^T @ T
// (reference pointnet.py:140-143), write xs (and the untransformed copy xo if non-null).
__device__ __forceinline__ void stage_points(const float *__restrict__ xb, int N, int tile, bool has_t,
                                             const float (&tm)[9], float *xs, float *xo, int tid) {
    if (tid < TP) {
        int n = tile * TP + tid;
        n = n < N ? n : N - 1;
        float x0 = xb[n], x1 = xb[N + n], x2 = xb[2 * N + n];
        if (xo) { xo[tid] = x0; xo[TP + tid] = x1; xo[2 * TP + tid] = x2; }
        if (has_t) {
            float y0 = fmaf(x2, tm[6], fmaf(x1, tm[3], x0 * tm[0]));
            float y1 = fmaf(x2, tm[7], fmaf(x1, tm[4], x0 * tm[1]));
            float y2 = fmaf(x2, tm[8], fmaf(x1, tm[5], x0 * tm[2]));
            x0 = y0; x1 = y1; x2 = y2;
        }
        xs[tid] = x0; xs[TP + tid] = x1; xs[2 * TP + tid] = x2;
    }
}

// Layer 1 (3 -> 64) on the VALU.  thread = (channel c = lane, 16-point group = wave): the lane's six per-channel
// constants stay in registers for the whole kernel (L1C, loaded once) and the tile's coordinates are LDS broadcast
// reads.  (The previous mapping — point per lane, 16 wave-uniform channels per wave — re-fetched 96 constants through
// the scalar cache for every tile: 8,100 of pass E's 33,500 cycles per wave and tile, tools/phase_times.py.)
//   z = w1[c]·x + b1[c];  h1 = relu(z * sc[c] + sh[c])      (sc == nullptr: h1 = relu(z))
// Same fmaf chain per element as before, so recomputed activations stay bit-identical across passes.
struct L1C { float w0, w1, w2, b, sc, sh; bool affine; };

__device__ __forceinline__ L1C load_l1c(const float *__restrict__ w1, const float *__restrict__ b1,
                                        const float *__restrict__ sc, const float *__restrict__ sh, const Lane &L) {
    L1C k;
    const int c = L.lane;
    k.w0 = w1[c * 3]; k.w1 = w1[c * 3 + 1]; k.w2 = w1[c * 3 + 2]; k.b = b1[c];
    k.affine = sc != nullptr;
    k.sc = k.affine ? sc[c] : 1.f;
    k.sh = k.affine ? sh[c] : 0.f;
    return k;
}

// (The affine / plain variants are two loops: a per-element select on k.affine costs a VALU instruction per activation.
// Packed FMAs on point pairs were measured and are not faster here — four dependent v_pk_fma_f32 per pair with a wait
// state each against two interleaved scalar chains: pass B +3 %.)
template <bool AFF>
__device__ __forceinline__ void layer1_rows(const float *xs, const L1C &k, float *dst, int p0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // 12 broadcast ds_read_b128: the wave's 16 points
        const f32x4 x0 = *(const f32x4 *)(xs + p0 + 4 * q);
        const f32x4 x1 = *(const f32x4 *)(xs + TP + p0 + 4 * q);
        const f32x4 x2 = *(const f32x4 *)(xs + 2 * TP + p0 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float z = fmaf(k.w2, x2[e], fmaf(k.w1, x1[e], fmaf(k.w0, x0[e], k.b)));
            if (AFF) z = fmaf(z, k.sc, k.sh);
            dst[(4 * q + e) * H1S] = fmaxf(z, 0.f);
        }
    }
}

__device__ __forceinline__ void layer1_tile(const float *xs, const L1C &k, float *h1, const Lane &L) {
    const int p0 = L.wave * 16;
    float *dst = h1 + p0 * H1S + L.lane;
    if (k.affine) layer1_rows<true>(xs, k, dst, p0);
    else layer1_rows<false>(xs, k, dst, p0);
}

// Layer 2 (64 -> 128) raw product for channel block cb (32 channels), both 32-point blocks:
//   acc{0,1}[r] = sum_k h1[point][k] * W2[cb*32 + j][k],  point = pb*32 + mfma_row(r, lane)
// w2p is the MFMA_B-packed (128,64) weight.
__device__ __forceinline__ void layer2_mfma(const float *h1, const float *__restrict__ w2p, int cb,
                                            const Lane &L, f32x16 &acc0, f32x16 &acc1) {
    const f32x4 *wp = (const f32x4 *)w2p + (size_t)(cb * 8) * 64 + L.lane;
    const float *a0p = h1 + L.j * H1S + L.h * 4;
    const float *a1p = h1 + (32 + L.j) * H1S + L.h * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll 4
    for (int kb = 0; kb < 8; ++kb) {
        f32x4 wv = wp[kb * 64];
        f32x4 a0 = *(const f32x4 *)(a0p + kb * 8);
        f32x4 a1 = *(const f32x4 *)(a1p + kb * 8);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc0 = mfma32(a0[t], wv[t], acc0);
            acc1 = mfma32(a1[t], wv[t], acc1);
        }
    }
}

// K = 128 product against an MFMA_B-packed (C,128) matrix, channel block cb, A rows from an
// h2-shaped LDS tile ([64][H2S]):  acc{0,1}[r] = sum_k tile[point][k] * Wp[cb*32 + j][k].
__device__ __forceinline__ void k128_mfma(const float *tile, const float *__restrict__ wp128, int cb,
                                          const Lane &L, f32x16 &acc0, f32x16 &acc1) {
    const f32x4 *wp = (const f32x4 *)wp128 + (size_t)(cb * 16) * 64 + L.lane;
    f32x4 wf[16];
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) wf[kb] = wp[kb * 64];
    const float *a0p = tile + L.j * H2S + L.h * 4;
    const float *a1p = tile + (32 + L.j) * H2S + L.h * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        f32x4 a0 = *(const f32x4 *)(a0p + kb * 8);
        f32x4 a1 = *(const f32x4 *)(a1p + kb * 8);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc0 = mfma32(a0[t], wf[kb][t], acc0);
            acc1 = mfma32(a1[t], wf[kb][t], acc1);
        }
    }
}

// The same product split in two so the caller can software-pipeline the weight fragments:
// load_wfrag() issues the 16 coalesced 1-KiB fragment loads of channel block cb, k128_compute()
// consumes a fragment set that is already (being) loaded.
__device__ __forceinline__ void load_wfrag(f32x4 (&wf)[16], const float *__restrict__ wp128, int cb, const Lane &L) {
    const f32x4 *wp = (const f32x4 *)wp128 + (size_t)(cb * 16) * 64 + L.lane;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) wf[kb] = wp[kb * 64];
}

__device__ __forceinline__ void k128_compute(const float *tile, const f32x4 (&wf)[16], const Lane &L,
                                             f32x16 &acc0, f32x16 &acc1) {
    const float *a0p = tile + L.j * H2S + L.h * 4;
    const float *a1p = tile + (32 + L.j) * H2S + L.h * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        f32x4 a0 = *(const f32x4 *)(a0p + kb * 8);
        f32x4 a1 = *(const f32x4 *)(a1p + kb * 8);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc0 = mfma32(a0[t], wf[kb][t], acc0);
            acc1 = mfma32(a1[t], wf[kb][t], acc1);
        }
    }
}

// Alternative tile layout (inference trunk, training pass C): LDS tiles UNPADDED and XOR-swizzled at float4 granularity instead of using the
// +4-float row pad of pngpd_tile.h: element (row, col) lives at row*W + (((col>>2) ^ (row&15))<<2) + (col&3).
// A wave's ds_read_b128 A-fragment read (lane i -> row i, one float4 column) then hits 16 distinct 16-B
// slots per 16-lane group (conflict-free) and the footprint drops to 54,016 B -> 3 workgroups per CU.
#define I1S 64
#define I2S 128

__device__ __forceinline__ int swz(int row, int col, int width) {
    return row * width + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3));
}

// layer1_tile() for the swizzled h1 tile: lane = channel, so for a (wave-uniform) point p the 64 lanes write the 64
// distinct floats of row p — conflict-free under the XOR swizzle as well.
__device__ __forceinline__ void layer1_tile_swz(const float *xs, const L1C &k, float *h1, const Lane &L) {
    const int p0 = L.wave * 16;
    const int c = L.lane, c4 = c >> 2, ce = c & 3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 x0 = *(const f32x4 *)(xs + p0 + 4 * q);
        const f32x4 x1 = *(const f32x4 *)(xs + TP + p0 + 4 * q);
        const f32x4 x2 = *(const f32x4 *)(xs + 2 * TP + p0 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int p = p0 + 4 * q + e;
            float z = fmaf(k.w2, x2[e], fmaf(k.w1, x1[e], fmaf(k.w0, x0[e], k.b)));
            if (k.affine) z = fmaf(z, k.sc, k.sh);
            h1[p * I1S + (((c4 ^ (p & 15)) << 2) | ce)] = fmaxf(z, 0.f);
        }
    }
}

// K-contraction of a swizzled [64][W] tile against register-resident MFMA_B fragments (NKB k-blocks).
template <int W, int NKB>
__device__ __forceinline__ void swz_compute(const float *tile, const f32x4 (&wf)[NKB], const Lane &L,
                                            f32x16 &acc0, f32x16 &acc1) {
    const float *r0 = tile + L.j * W, *r1 = tile + (32 + L.j) * W;
    const int sx = L.j & 15;   // (32 + j) & 15 == j & 15
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // A fragments double-buffered in registers: the two ds_read_b128 of k-block kb+1 are in flight while the
    // eight MFMAs of k-block kb issue (hipcc otherwise re-uses the registers and waits lgkmcnt(0) every block)
    f32x4 a0 = *(const f32x4 *)(r0 + (((0 * 2 + L.h) ^ sx) << 2));
    f32x4 a1 = *(const f32x4 *)(r1 + (((0 * 2 + L.h) ^ sx) << 2));
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        f32x4 n0 = a0, n1 = a1;
        if (kb + 1 < NKB) {
            const int c4 = (((kb + 1) * 2 + L.h) ^ sx) << 2;
            n0 = *(const f32x4 *)(r0 + c4);
            n1 = *(const f32x4 *)(r1 + c4);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc0 = mfma32(a0[t], wf[kb][t], acc0);
            acc1 = mfma32(a1[t], wf[kb][t], acc1);
        }
        a0 = n0; a1 = n1;
    }
}


// ---- software-pipelined variants for the padded layout (training passes B, D, E, gather) -------------------
// Layer-2 weight fragments of channel block cb (8 coalesced 1-KiB loads); issue early, consume after the barrier.
__device__ __forceinline__ void load_w2frag(f32x4 (&wf)[8], const float *__restrict__ w2p, int cb, const Lane &L) {
    const f32x4 *wp = (const f32x4 *)w2p + (size_t)(cb * 8) * 64 + L.lane;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) wf[kb] = wp[kb * 64];
}

// Layer 2 on register-resident weights with the A fragments double-buffered (the ds_read_b128 pair of k-block
// kb+1 is in flight while the eight MFMAs of k-block kb issue).
__device__ __forceinline__ void layer2_compute(const float *h1, const f32x4 (&wf)[8], const Lane &L,
                                               f32x16 &acc0, f32x16 &acc1) {
    const float *a0p = h1 + L.j * H1S + L.h * 4;
    const float *a1p = h1 + (32 + L.j) * H1S + L.h * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    f32x4 a0 = *(const f32x4 *)a0p, a1 = *(const f32x4 *)a1p;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        f32x4 n0 = a0, n1 = a1;
        if (kb + 1 < 8) { n0 = *(const f32x4 *)(a0p + (kb + 1) * 8); n1 = *(const f32x4 *)(a1p + (kb + 1) * 8); }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc0 = mfma32(a0[t], wf[kb][t], acc0);
            acc1 = mfma32(a1[t], wf[kb][t], acc1);
        }
        a0 = n0; a1 = n1;
    }
}

// K = 128 product accumulated INTO acc0/acc1 (not zeroed here), weights streamed from L2 through a RING-deep register
// ring (fragment kb+RING is requested when fragment kb is consumed: RING x 256 (NPB = 1) or x 512 (NPB = 2) MFMA cycles
// of cover — an L2 round trip under load is 1,000-2,000 cycles) and the A fragments double-buffered.  Both point blocks (NPB = 2) or only block pb (NPB = 1, acc1 unused).
template <int NPB, int RING = 4>
__device__ __forceinline__ void k128_stream(const float *tile, const float *__restrict__ wp128, int cb, int pb,
                                            const Lane &L, f32x16 &acc0, f32x16 &acc1) {
    const f32x4 *wp = (const f32x4 *)wp128 + (size_t)(cb * 16) * 64 + L.lane;
    const float *a0p = tile + (pb * 32 + L.j) * H2S + L.h * 4;
    const float *a1p = tile + (32 + L.j) * H2S + L.h * 4;
    f32x4 wq[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) wq[i] = wp[i * 64];
    f32x4 a0 = *(const f32x4 *)a0p, a1 = a0;
    if (NPB == 2) a1 = *(const f32x4 *)a1p;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        f32x4 n0 = a0, n1 = a1;
        if (kb + 1 < 16) {
            n0 = *(const f32x4 *)(a0p + (kb + 1) * 8);
            if (NPB == 2) n1 = *(const f32x4 *)(a1p + (kb + 1) * 8);
        }
        const f32x4 wv = wq[kb % RING];
        if (kb + RING < 16) wq[kb % RING] = wp[(kb + RING) * 64];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc0 = mfma32(a0[t], wv[t], acc0);
            if (NPB == 2) acc1 = mfma32(a1[t], wv[t], acc1);
        }
        a0 = n0; a1 = n1;
    }
}

// The same product (one point block, NPB = 1) with the weight fragments of k-blocks [0, K128_LDS_KB) resident in LDS
// (wl: [kb][cb][lane] f32x4, filled once per workgroup by k128_fill_lds) and only the last 16 - K128_LDS_KB streamed
// from L2 — requested first, consumed last, so K128_LDS_KB x 256 MFMA cycles cover their round trip.  With one point
// block a streamed fragment buys only 256 MFMA cycles against a 1,000-2,000-cycle L2 round trip: pass E's first
// contraction spent 15.3 k cycles per tile on 4.1 k cycles of matrix work (profiles/r03_phase_times.txt).  Same MFMA
// order as k128_stream (kb, t ascending): bit-identical.
#define K128_LDS_KB 13
#define K128_LDS_FLOATS (K128_LDS_KB * 2 * 64 * 4)
__device__ __forceinline__ void k128_fill_lds(float *wl, const float *__restrict__ wp128, int tid) {
    const f32x4 *src = (const f32x4 *)wp128;
    f32x4 *dst = (f32x4 *)wl;
    for (int e = tid; e < K128_LDS_KB * 2 * 64; e += 256) {
        const int lane = e & 63, cb = (e >> 6) & 1, kb = e >> 7;
        dst[e] = src[(size_t)(cb * 16 + kb) * 64 + lane];
    }
}
__device__ __forceinline__ void k128_lds(const float *tile, const float *wl, const float *__restrict__ wp128, int cb,
                                         int pb, const Lane &L, f32x16 &acc0) {
    constexpr int NS = 16 - K128_LDS_KB;
    const f32x4 *wp = (const f32x4 *)wp128 + (size_t)(cb * 16) * 64 + L.lane;
    const f32x4 *wlp = (const f32x4 *)wl + cb * 64 + L.lane;
    const float *a0p = tile + (pb * 32 + L.j) * H2S + L.h * 4;
    f32x4 wq[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) wq[i] = wp[(K128_LDS_KB + i) * 64];
    f32x4 a0 = *(const f32x4 *)a0p, w0 = wlp[0];
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        f32x4 n0 = a0, nw = w0;
        if (kb + 1 < 16) n0 = *(const f32x4 *)(a0p + (kb + 1) * 8);
        if (kb + 1 < K128_LDS_KB) nw = wlp[(kb + 1) * 128];
        const f32x4 wv = kb < K128_LDS_KB ? w0 : wq[kb < K128_LDS_KB ? 0 : kb - K128_LDS_KB];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc0 = mfma32(a0[t], wv[t], acc0);
        a0 = n0; w0 = nw;
    }
}

// ---- bf16 matrix-core variants (opt-in reduced precision, pngpd_bf.h): the LDS tiles stay fp32, operands are
// converted (NT = 1) or split hi/lo (NT = 3) when they are read — the same LDS traffic as the fp32 loops, an eighth of
// the MFMA issue slots per product term.
// Layer 2 (64 -> 128) for channel block cb: A rows from the fp32 h1 tile, B from split_pack_bf16 fragments (KS = 4).
template <int NT>
__device__ __forceinline__ void layer2_compute_bf(const float *h1, const u16 *__restrict__ w2x, int cb, const Lane &L,
                                                  f32x16 &acc0, f32x16 &acc1) {
    const float *a0p = h1 + L.j * H1S + L.h * 8;
    const float *a1p = h1 + (32 + L.j) * H1S + L.h * 8;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        f32x4 wh, wl, ah, al;
        bf_wfrag<NT>(w2x, 4, cb, ks, L.lane, wh, wl);
        bf_pack8<NT>(*(const f32x4 *)(a0p + ks * 16), *(const f32x4 *)(a0p + ks * 16 + 4), ah, al);
        acc0 = bf_mma<NT>(ah, al, wh, wl, acc0);
        bf_pack8<NT>(*(const f32x4 *)(a1p + ks * 16), *(const f32x4 *)(a1p + ks * 16 + 4), ah, al);
        acc1 = bf_mma<NT>(ah, al, wh, wl, acc1);
    }
}

// K = 128 product of rows [pb*32, pb*32+32) of an h2-shaped fp32 tile against split_pack_bf16 fragments of a (C,128)
// matrix, channel block cb, accumulated INTO acc.
template <int NT>
__device__ __forceinline__ void k128_bf(const float *tile, const u16 *__restrict__ wx, int cb, int pb, const Lane &L,
                                        f32x16 &acc) {
    const float *ap = tile + (pb * 32 + L.j) * H2S + L.h * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        f32x4 wh, wl, ah, al;
        bf_wfrag<NT>(wx, 8, cb, ks, L.lane, wh, wl);
        bf_pack8<NT>(*(const f32x4 *)(ap + ks * 16), *(const f32x4 *)(ap + ks * 16 + 4), ah, al);
        acc = bf_mma<NT>(ah, al, wh, wl, acc);
    }
}

// Both point blocks of the tile against the same fragments (pass D's h2 A): each fragment pair is fetched once and
// serves two accumulators (two separate k128_bf calls fetched it twice); per accumulator the order is k128_bf's.
template <int NT>
__device__ __forceinline__ void k128_bf2(const float *tile, const u16 *__restrict__ wx, int cb, const Lane &L,
                                         f32x16 &acc0, f32x16 &acc1) {
    const float *a0p = tile + L.j * H2S + L.h * 8;
    const float *a1p = tile + (32 + L.j) * H2S + L.h * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        f32x4 wh, wl, ah, al;
        bf_wfrag<NT>(wx, 8, cb, ks, L.lane, wh, wl);
        bf_pack8<NT>(*(const f32x4 *)(a0p + ks * 16), *(const f32x4 *)(a0p + ks * 16 + 4), ah, al);
        acc0 = bf_mma<NT>(ah, al, wh, wl, acc0);
        bf_pack8<NT>(*(const f32x4 *)(a1p + ks * 16), *(const f32x4 *)(a1p + ks * 16 + 4), ah, al);
        acc1 = bf_mma<NT>(ah, al, wh, wl, acc1);
    }
}

// The same with the fragments of k-steps [0, K128BF_LDS_KS) resident in LDS (wl: [ks][cb][part][lane] quads, hi only
// for NT == 1; filled once per workgroup) and the rest streamed — requested first, consumed last.  Pass E's first
// contraction covers one point block per wave: a streamed fragment pair buys 3 (1) bf16 matrix instructions against an L2
// round trip.  Same instruction order as k128_bf: bit-identical.
#define K128BF_LDS_KS(NT) ((NT) == 3 ? 6 : 8)      // 24 KB (hi + lo) / 16 KB (hi): within K128_LDS_FLOATS
template <int NT>
__device__ __forceinline__ void k128_bf_fill_lds(float *wl, const u16 *__restrict__ wx, int tid) {
    constexpr int NK = K128BF_LDS_KS(NT), NP = NT == 3 ? 2 : 1;
    const f32x4 *src = (const f32x4 *)wx;
    f32x4 *dst = (f32x4 *)wl;
    for (int e = tid; e < NK * 2 * NP * 64; e += 256) {
        const int lane = e & 63, q = e >> 6;
        const int part = q % NP, cb = (q / NP) & 1, ks = q / (2 * NP);
        dst[e] = src[((size_t)(cb * 8 + ks) * 2 + part) * 64 + lane];
    }
}
template <int NT>
__device__ __forceinline__ void k128_bf_lds(const float *tile, const float *wl, const u16 *__restrict__ wx, int cb,
                                            int pb, const Lane &L, f32x16 &acc) {
    constexpr int NK = K128BF_LDS_KS(NT), NP = NT == 3 ? 2 : 1, NS = 8 - NK;
    const float *ap = tile + (pb * 32 + L.j) * H2S + L.h * 8;
    const f32x4 *wq = (const f32x4 *)wl + (size_t)cb * NP * 64 + L.lane;
    f32x4 sh[NS > 0 ? NS : 1], sl[NS > 0 ? NS : 1];
#pragma unroll
    for (int i = 0; i < NS; ++i) bf_wfrag<NT>(wx, 8, cb, NK + i, L.lane, sh[i], sl[i]);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        f32x4 wh, wlo, ah, al;
        if (ks < NK) {
            wh = wq[(size_t)ks * 2 * NP * 64];
            if (NT == 3) wlo = wq[(size_t)ks * 2 * NP * 64 + 64]; else wlo = wh;
        } else {
            wh = sh[ks < NK ? 0 : ks - NK]; wlo = sl[ks < NK ? 0 : ks - NK];
        }
        bf_pack8<NT>(*(const f32x4 *)(ap + ks * 16), *(const f32x4 *)(ap + ks * 16 + 4), ah, al);
        acc = bf_mma<NT>(ah, al, wh, wlo, acc);
    }
}

// Experiment hook (variant builds: -DPNGPD_PRIO=n): a static priority for every other resident workgroup of a CU.
#ifndef PNGPD_PRIO
#define PNGPD_PRIO 0
#endif
#ifndef PNGPD_PRIO_SHIFT
#define PNGPD_PRIO_SHIFT 8
#endif
__device__ __forceinline__ void wg_priority() {
#if PNGPD_PRIO
    if ((blockIdx.x >> PNGPD_PRIO_SHIFT) & 1) __builtin_amdgcn_s_setprio(PNGPD_PRIO);
#endif
}

// Split of a cloud's T tiles over S workgroups.
__device__ __forceinline__ void tile_range(int s, int S, int T, int &t0, int &t1) {
    t0 = (int)(((long)s * T) / S);
    t1 = (int)(((long)(s + 1) * T) / S);
}
