// libpngpd — the two fp64 finalize steps of the trunk backward as DEVICE FUNCTIONS:
//   dw3_finalize_body    dW3[c][j] = G[c][j] - s3 (m1 sh[j] + (m2/sig3) (W3 Sc)[c][j]),  Sc = S2 - sh sh^T / M
//   a_cvec_finalize_body A = W3^T diag(g3 m2/sig3^2) W3 (MFMA_B-packed fp32),  cvec = A mh - W3^T (s3 m1)
// Each is a chain of a few L2 round trips on 128-256 workgroups — 14 and 10 us as launches of their own, on a mostly
// idle chip.  Neither has a consumer inside the pass that follows its inputs (dW3 is read by the optimizer; A / cvec
// by pass D, one launch later), so in the fused backward they ride as TAIL WORKGROUPS of that pass: dW3's finalize
// behind pass E's grid, A / cvec's behind the gather pass's (pngpd_train.hip) — independent work in one launch, no
// hand-off between workgroups.  The kernels of pngpd_train_glue.hip (per-op entries) call the same bodies: same
// arithmetic in the same order for every workgroup size, so the two sequencings stay bit-identical.
#pragma once
#include "pngpd_common.h"

// Sum over the 64 lanes of a wave (butterfly: every lane gets the same, order-fixed total).
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// ---------------------------------------------------------------------------------------
// dW3: block = DW3_CPB channels c.  A thread walks column j of S2 block-wise (the storage orientation of a 32x32 block is
// decided once per block, not per element) and every element it fetches serves all DW3_CPB channels of the block.
// 128 x KQ threads: KQ row blocks a (k = 32 a + i) are walked concurrently, the rest in turn; the four partial dot
// products of a column meet in LDS in fixed order, so KQ does not change the result.  One barrier before the walk and
// one after it; the 128-long dot products w3 . sh go through wave shuffles beside the walk; all 32 (64) loads of a
// thread's S2 block are in flight at once.
// S2c f64 [12][16][64]: the 10 accumulator blocks pass D keeps (slot 3w+q of wave w = block (w, (w+q) mod 4) for q < 2;
// slot 3w+2 = the partial of block (w, w+2) [w < 2] or of block (w-2, w) [w >= 2] over half of each tile's points — the
// two are added here; raw MFMA register layout); the other six blocks are transposes.
// ---------------------------------------------------------------------------------------
#define DW3_CPB 4
#define DW3_KQ (PNGPD_ASAN ? 1 : 4)   // the stand-alone kernel's k-quarters (sanitizer build: 128-thread workgroups)
#define DW3_LDS_DOUBLES (DW3_CPB * 128 + DW3_CPB * 2 + 4 * DW3_CPB * 128)
struct DW3Args {
    const double *G, *S2c, *sh; double M;
    const float *w3, *g3; const double *stats, *m12; double eps; float *dW3;
};

template <int KQ>
__device__ __forceinline__ void dw3_finalize_body(const DW3Args &A, int blk, double *lds) {
    double (*wrow)[128] = (double (*)[128])lds;                                      // [DW3_CPB][128]
    double (*rsum)[2] = (double (*)[2])(lds + DW3_CPB * 128);                        // [DW3_CPB][2]
    double (*part)[DW3_CPB][128] = (double (*)[DW3_CPB][128])(lds + DW3_CPB * 130);  // [4][DW3_CPB][128]
    const int c0 = blk * DW3_CPB, j = threadIdx.x & 127, aq = threadIdx.x >> 7;
    const double shj = A.sh[j];
    for (int u = aq; u < DW3_CPB; u += KQ) {
        const double w = (double)A.w3[(size_t)(c0 + u) * 128 + j];
        wrow[u][j] = w;
        const double t = wave_sum_f64(w * shj);
        if ((threadIdx.x & 63) == 0) rsum[u][j >> 6] = t;
    }
    double gq[DW3_CPB];
    if (aq == 0) {
#pragma unroll
        for (int u = 0; u < DW3_CPB; ++u) gq[u] = A.G[(size_t)(c0 + u) * 128 + j];
    }
    __syncthreads();
    const int bb = j >> 5, jj = j & 31;
    for (int a = aq; a < 4; a += KQ) {          // rows k = 32 a + i of column j: block (a, bb) of S2
        double dot[DW3_CPB];
#pragma unroll
        for (int u = 0; u < DW3_CPB; ++u) dot[u] = 0.0;
        const int d = (bb - a) & 3;
        const bool tr = d == 3 || (d == 2 && a >= 2);          // stored as the transposed block (bb, a)
        const int ra = tr ? bb : a, q = tr ? ((a - bb) & 3) : d;
        const double *p0 = A.S2c + (size_t)(ra * 3 + q) * 1024;
        const double *p1 = A.S2c + (size_t)((q == 2 ? ra + 2 : ra) * 3 + 2) * 1024;   // q == 2: the other half of the points
        double v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            // stored element (row, col) at ((row&3) + 4 (row>>3)) * 64 + ((row>>2)&1) * 32 + col
            const int row = tr ? jj : i, col = tr ? i : jj;
            const int e = ((row & 3) + 4 * (row >> 3)) * 64 + ((row >> 2) & 1) * 32 + col;
            v[i] = p0[e];
            if (q == 2) v[i] += p1[e];
        }
#pragma unroll
        for (int i = 0; i < 32; ++i)
#pragma unroll
            for (int u = 0; u < DW3_CPB; ++u) dot[u] = fma(wrow[u][a * 32 + i], v[i], dot[u]);
#pragma unroll
        for (int u = 0; u < DW3_CPB; ++u) part[a][u][j] = dot[u];
    }
    __syncthreads();
    if (aq == 0) {
#pragma unroll
        for (int u = 0; u < DW3_CPB; ++u) {
            const int c = c0 + u;
            const double w3sc = ((part[0][u][j] + part[1][u][j]) + (part[2][u][j] + part[3][u][j])) -
                                (rsum[u][0] + rsum[u][1]) * shj / A.M;
            const double sig = sqrt(A.stats[1024 + c] + A.eps);
            const double s3 = (double)A.g3[c] / sig;
            A.dW3[(size_t)c * 128 + j] = (float)(gq[u] - s3 * (A.m12[c] * shj + (A.m12[1024 + c] / sig) * w3sc));
        }
    }
}

// ---------------------------------------------------------------------------------------
// A / cvec: block = row i of A; 512 LOGICAL threads = 128 columns j x 4 quarters of the channel range.  The per-channel
// coefficient (one fp64 divide + sqrt each) is computed ONCE per block into LDS, so the contraction loop is one
// coalesced weight load + one LDS broadcast + one FMA per channel, 16 loads in flight.  NTH = 512: one logical thread
// per thread; NTH = 256 (tail workgroups of the gather pass): thread t also plays logical thread t + 256 — the same 512
// partial sums, the same trees, the same result.
// ---------------------------------------------------------------------------------------
#define ACVEC_LDS_DOUBLES (1024 + 4 * 128 + 512 + 128)
struct ACvecArgs {
    const float *w3, *g3; const double *stats, *m12, *sh; double M, eps; float *Ap, *cvec;
};

template <int NTH>
__device__ __forceinline__ void a_cvec_finalize_body(const ACvecArgs &A, int i, double *lds) {
    static_assert(NTH == 512 || NTH == 256, "512 logical threads on 512 or 256 physical ones");
    double *tco = lds;                                   // [1024]  w3[c][i] * g3[c] m2[c] / var[c]
    double (*part)[128] = (double (*)[128])(lds + 1024); // [4][128]
    double *ured = lds + 1024 + 512;                     // [512]
    double *red = ured + 512;                            // [128]
    const int tid = threadIdx.x;
#pragma unroll
    for (int lt = tid; lt < 512; lt += NTH) {            // logical thread lt: channels lt and lt + 512
        double u = 0.0;
        for (int c = lt; c < 1024; c += 512) {
            const double var = A.stats[1024 + c] + A.eps;
            const double wi = (double)A.w3[(size_t)c * 128 + i];
            tco[c] = wi * ((double)A.g3[c] * A.m12[1024 + c] / var);
            u += wi * ((double)A.g3[c] / sqrt(var)) * A.m12[c];
        }
        ured[lt] = u;
    }
    __syncthreads();
#pragma unroll
    for (int lt = tid; lt < 512; lt += NTH) {
        const int j = lt & 127, q = lt >> 7;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        const float *wj = A.w3 + (size_t)(q * 256) * 128 + j;
        const double *tq = tco + q * 256;
#pragma unroll 4
        for (int c = 0; c < 256; c += 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += tq[c + e] * (double)wj[(size_t)(c + e) * 128];
        }
        part[q][j] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    for (int s = 256; s > 0; s >>= 1) {
        __syncthreads();
        for (int lt = tid; lt < s; lt += NTH) ured[lt] += ured[lt + s];
    }
    __syncthreads();
    if (tid < 128) {
        const int j = tid;
        const double a = part[0][j] + part[1][j] + part[2][j] + part[3][j];
        const int cb = i >> 5, jj = i & 31, kb = j >> 3, h = (j >> 2) & 1, t = j & 3;   // MFMA_B packing of (i, j)
        A.Ap[(((cb * 16 + kb) * 64) + h * 32 + jj) * 4 + t] = (float)a;
        red[j] = a * (A.sh[j] / A.M);
    }
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    if (tid == 0) A.cvec[i] = (float)(red[0] - ured[0]);
}
