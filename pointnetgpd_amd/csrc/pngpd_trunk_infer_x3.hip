// libpngpd — OPT-IN fast inference trunk: the same fused per-point MLP + max-pool as pngpd_trunk_infer.hip,
// with the two GEMM layers evaluated by "bf16x3" split arithmetic on the bf16 matrix cores:
//     a = a_hi + a_lo,  w = w_hi + w_lo  (each part bf16, hi = rne(a), lo = rne(a - hi))
//     a*w ~= a_hi*w_hi + a_hi*w_lo + a_lo*w_hi        (products exact in fp32, fp32 accumulate)
// i.e. 3 v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate each) per fp32 product block.  The dropped
// a_lo*w_lo term and the 16-bit split bound the relative error of every product by ~2^-16 (1.5e-5):
// log-probabilities stay within 1e-4 of the exact-fp32 path (tests), inside the 1e-3 contract, but the
// result is NOT bit-identical to fp32 arithmetic — hence opt-in (precision="bf16x3").
//
// One workgroup = 512 threads = 8 waves (2 per SIMD), one cloud, tiles of 128 points.
//   LDS: h1 hi/lo [128][72] bf16, h2 hi/lo [128][136] bf16 (+8 halfword row pad: conflict-free
//   ds_read_b128), xs [3][128] f32, running max [1024] f32  ->  ~112 KB, one workgroup per CU.
#include "pngpd_bf.h"

#ifdef PNGPD_TIMING
__device__ unsigned long long pngpd_tm_x3[16];
extern "C" int pngpd_tm_read_x3(unsigned long long *host16, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host16, HIP_SYMBOL(pngpd_tm_x3), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(pngpd_tm_x3), z, sizeof(z)); }
    return 0;
}
#endif

#define XP 128          // points per tile
#define X1S 72          // h1 row stride (halfwords)
#define X2S 136         // h2 row stride (halfwords)
#define X3_LDS_BYTES (2 * XP * X1S * 2 + 2 * XP * X2S * 2 + 3 * XP * 4 + 1024 * 4)

// (C,K) fp32 row-major -> hi/lo bf16 in 32x32x16 B-fragment order:
//   out[(((cb*KS + ks)*2 + part)*64 + lane)*8 + t] = part(W[cb*32 + (lane&31)][ks*16 + (lane>>5)*8 + t])
__global__ void split_pack_bf16_kernel(const float *__restrict__ W, int C, int K, u16 *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * K) return;
    const int c = idx / K, k = idx - c * K;
    u16 hi, lo;
    split2(W[idx], hi, lo);
    const int cb = c >> 5, j = c & 31, ks = k >> 4, h = (k >> 3) & 1, t = k & 7, KS = K >> 4;
    const size_t base = ((size_t)(cb * KS + ks) * 2) * 64 * 8 + (size_t)(h * 32 + j) * 8 + t;
    out[base] = hi;
    out[base + 64 * 8] = lo;
}

// NT = number of product terms: 3 = bf16x3 (hi*hi + hi*lo + lo*hi), 1 = plain bf16 (hi*hi only; the lo halves of
// the packed weights and the lo activation tiles are neither loaded nor written).
template <int KS, int NT>
__device__ __forceinline__ void load_wx(f32x4 (&wh)[KS], f32x4 (&wl)[KS], const u16 *__restrict__ wx, int cb, int lane) {
    const f32x4 *p = (const f32x4 *)wx + (size_t)(cb * KS) * 2 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        wh[ks] = p[(ks * 2) * 64];
        if (NT == 3) wl[ks] = p[(ks * 2 + 1) * 64]; else wl[ks] = wh[ks];
    }
}

// one cloud coordinate from fp32 or bf16 storage
template <bool XBF>
__device__ __forceinline__ float ldx(const void *__restrict__ base, size_t i) {
    return XBF ? bf16_val(((const u16 *)base)[i]) : ((const float *)base)[i];
}

// ARG: also track WHICH point gave every channel's maximum (out_arg, one int per (cloud, split, channel)) — the input of
// the pool refinement (pngpd_trunk_pool_refine): the reduced-precision pass then only CHOOSES the point, its value is
// re-evaluated in exact fp32.  The arg search (64 compare / select pairs per block) runs only in blocks where some lane
// of the wave saw a new maximum.
template <int NT, bool XBF, bool ARG>
__global__ __launch_bounds__(512, 2) void trunk_infer_x3_kernel(
    const void *__restrict__ x, int N, const float *__restrict__ trans,
    const float *__restrict__ w1, const float *__restrict__ b1,
    const u16 *__restrict__ w2x, const float *__restrict__ b2,
    const u16 *__restrict__ w3x, const float *__restrict__ b3,
    int relu_last, int T, int S, float *__restrict__ out, int *__restrict__ out_arg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16 *h1h = (u16 *)smem_raw;                 // [XP][X1S]
    u16 *h1l = h1h + XP * X1S;
    u16 *h2h = h1l + XP * X1S;                  // [XP][X2S]
    u16 *h2l = h2h + XP * X2S;
    float *xs = (float *)(h2l + XP * X2S);      // [3][XP]
    float *rm = xs + 3 * XP;                    // [1024]
    int *ri = (int *)(rm + 1024);               // [1024] running arg-max (ARG only; the launch adds the 4 KB)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / S, s = blockIdx.x - b * S;
    const int t0 = (int)(((long)s * T) / S), t1 = (int)(((long)(s + 1) * T) / S);
    const size_t xo = (size_t)b * 3 * N;   // element offset of this cloud
    float tm[9] = {0};
    const bool has_t = trans != nullptr;
    if (has_t) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tm[i] = trans[(size_t)b * 9 + i];
    }
    for (int i = tid; i < 1024; i += 512) { rm[i] = -INFINITY; if (ARG) ri[i] = 0; }
    __shared__ int s_bad;   // non-finite input coordinate seen: poison the pooled row (see trunk_infer_kernel)
    if (tid == 0) s_bad = 0;
    __syncthreads();

    // layer-3 weight fragments (hi+lo of one 32-channel block = 64 VGPRs)
    f32x4 wah[8], wal[8];
    // NT == 1 (plain bf16): the wave's FOUR channel blocks stay resident in 128 VGPRs for the whole kernel.  With the
    // matrix cores 16x faster than in fp32 the weight stream itself was the bottleneck: every wave re-fetched 8 KB per
    // block and tile from L2 (2 KB per point and workgroup; with all 256 CUs on the same 256 KB of weights: half of each
    // XCD's L2 bandwidth), and tools/phase_times_x3.py showed 2,150 cycles per block waiting for the fragments to land —
    // 35 % of the kernel.  (NT == 3 would need 256 VGPRs for hi + lo of four blocks and keeps streaming.)
    constexpr bool WRES = NT == 1 && !ARG;   // (the arg search needs the registers: the ARG variant streams its weights)
    f32x4 wres[WRES ? 4 : 1][8];
    if (WRES) {
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            const f32x4 *p = (const f32x4 *)w3x + (size_t)((wave + 8 * ci) * 8) * 2 * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) wres[ci][ks] = p[(ks * 2) * 64];
        }
    }

    float px0 = 0.f, px1 = 0.f, px2 = 0.f;
    if (tid < XP) {
        int n = t0 * XP + tid; n = n < N ? n : N - 1;
        px0 = ldx<XBF>(x, xo + n); px1 = ldx<XBF>(x, xo + N + n); px2 = ldx<XBF>(x, xo + 2 * (size_t)N + n);
    }

    TM_DECL
    for (int tile = t0; tile < t1; ++tile) {
        TM(0)
        if (tid < XP) {
            float x0 = px0, x1 = px1, x2 = px2;
            if (has_t) {
                x0 = fmaf(px2, tm[6], fmaf(px1, tm[3], px0 * tm[0]));
                x1 = fmaf(px2, tm[7], fmaf(px1, tm[4], px0 * tm[1]));
                x2 = fmaf(px2, tm[8], fmaf(px1, tm[5], px0 * tm[2]));
            }
            xs[tid] = x0; xs[XP + tid] = x1; xs[2 * XP + tid] = x2;
            if (!__builtin_isfinite(px0 + px1 + px2)) s_bad = 1;
            if (tile + 1 < t1) {
                int n = (tile + 1) * XP + tid; n = n < N ? n : N - 1;
                px0 = ldx<XBF>(x, xo + n); px1 = ldx<XBF>(x, xo + N + n); px2 = ldx<XBF>(x, xo + 2 * (size_t)N + n);
            }
        }
        TM(1)
        __syncthreads();
        TM(2)
        {   // layer 1 (fp32 VALU): thread = (point p = tid & 127, 16-channel group g = tid >> 7 = wave >> 1)
            const int p = tid & 127, g = wave >> 1;
            const float x0 = xs[p], x1 = xs[XP + p], x2 = xs[2 * XP + p];
            float zv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = g * 16 + e;   // wave-uniform -> scalar loads
                zv[e] = fmaxf(fmaf(w1[c * 3 + 2], x2, fmaf(w1[c * 3 + 1], x1, fmaf(w1[c * 3], x0, b1[c]))), 0.f);
            }
            uint4 *dh = (uint4 *)(h1h + p * X1S + g * 16), *dl = (uint4 *)(h1l + p * X1S + g * 16);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                // channel pairs through ONE v_cvt_pk_bf16_f32 each (pngpd_bf.h bf_pk2 / bf_split_pk2): same bits as the
                // per-value split2 this replaced, a third of its instructions
                uint4 vh, vl = {0u, 0u, 0u, 0u};
                if (NT == 3) {
                    bf_split_pk2(zv[q * 8 + 0], zv[q * 8 + 1], vh.x, vl.x); bf_split_pk2(zv[q * 8 + 2], zv[q * 8 + 3], vh.y, vl.y);
                    bf_split_pk2(zv[q * 8 + 4], zv[q * 8 + 5], vh.z, vl.z); bf_split_pk2(zv[q * 8 + 6], zv[q * 8 + 7], vh.w, vl.w);
                } else {
                    vh.x = bf_pk2(zv[q * 8 + 0], zv[q * 8 + 1]); vh.y = bf_pk2(zv[q * 8 + 2], zv[q * 8 + 3]);
                    vh.z = bf_pk2(zv[q * 8 + 4], zv[q * 8 + 5]); vh.w = bf_pk2(zv[q * 8 + 6], zv[q * 8 + 7]);
                }
                dh[q] = vh;
                if (NT == 3) dl[q] = vl;
            }
        }
        TM(3)
        __syncthreads();
        TM(4)
        {   // layer 2 (64 -> 128), bf16x3: wave owns channel block cb = wave & 3 and point blocks 2q, 2q+1
            const int cb = wave & 3, pb0 = (wave >> 2) * 2;
            f32x4 w2h[4], w2l[4];
            load_wx<4, NT>(w2h, w2l, w2x, cb, lane);
            f32x16 a0 = {0}, a1 = {0};
            const int r0 = (pb0 * 32 + j) * X1S + h * 8, r1 = ((pb0 + 1) * 32 + j) * X1S + h * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 ah0 = *(const f32x4 *)(h1h + r0 + ks * 16), ah1 = *(const f32x4 *)(h1h + r1 + ks * 16);
                a0 = mfma_bf(ah0, w2h[ks], a0); a1 = mfma_bf(ah1, w2h[ks], a1);
                if (NT == 3) {
                    const f32x4 al0 = *(const f32x4 *)(h1l + r0 + ks * 16), al1 = *(const f32x4 *)(h1l + r1 + ks * 16);
                    a0 = mfma_bf(ah0, w2l[ks], a0); a1 = mfma_bf(ah1, w2l[ks], a1);
                    a0 = mfma_bf(al0, w2h[ks], a0); a1 = mfma_bf(al1, w2h[ks], a1);
                }
            }
            const float bias = b2[cb * 32 + j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma_row(r, lane);
                unsigned hi, lo = 0u;   // the register's two values (point blocks pb0, pb0 + 1) converted as a pair
                const float v0 = fmaxf(a0[r] + bias, 0.f), v1 = fmaxf(a1[r] + bias, 0.f);
                if (NT == 3) bf_split_pk2(v0, v1, hi, lo); else hi = bf_pk2(v0, v1);
                h2h[(pb0 * 32 + row) * X2S + cb * 32 + j] = (u16)hi;
                h2h[((pb0 + 1) * 32 + row) * X2S + cb * 32 + j] = (u16)(hi >> 16);
                if (NT == 3) {
                    h2l[(pb0 * 32 + row) * X2S + cb * 32 + j] = (u16)lo;
                    h2l[((pb0 + 1) * 32 + row) * X2S + cb * 32 + j] = (u16)(lo >> 16);
                }
            }
        }
        TM(5)
        __syncthreads();
        TM(6)
        // layer 3 (128 -> 1024), bf16x3: wave owns channel blocks wave + 8*ci, all four point blocks
        // One channel block x all four point blocks per step: 4 independent accumulator chains (an accumulator is
        // re-used every 4th MFMA) and the A fragments of k-step ks+1 in flight while k-step ks issues.
        auto block4 = [&](int cb, const f32x4 (&wah)[8], const f32x4 (&wal)[8]) {
            const float rmc = rm[cb * 32 + j];   // requested before the block's MFMAs, merged after them
            f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
            const int ro = j * X2S + h * 8;
            f32x4 ah[4], al[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ah[q] = *(const f32x4 *)(h2h + ro + q * 32 * X2S);
                al[q] = (NT == 3) ? *(const f32x4 *)(h2l + ro + q * 32 * X2S) : ah[q];
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                f32x4 nh[4], nl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    nh[q] = ah[q]; nl[q] = al[q];
                    if (ks < 7) {
                        nh[q] = *(const f32x4 *)(h2h + ro + q * 32 * X2S + (ks + 1) * 16);
                        if (NT == 3) nl[q] = *(const f32x4 *)(h2l + ro + q * 32 * X2S + (ks + 1) * 16);
                    }
                }
                c0 = mfma_bf(ah[0], wah[ks], c0); c1 = mfma_bf(ah[1], wah[ks], c1);
                c2 = mfma_bf(ah[2], wah[ks], c2); c3 = mfma_bf(ah[3], wah[ks], c3);
                if (NT == 3) {
                    c0 = mfma_bf(ah[0], wal[ks], c0); c1 = mfma_bf(ah[1], wal[ks], c1);
                    c2 = mfma_bf(ah[2], wal[ks], c2); c3 = mfma_bf(ah[3], wal[ks], c3);
                    c0 = mfma_bf(al[0], wah[ks], c0); c1 = mfma_bf(al[1], wah[ks], c1);
                    c2 = mfma_bf(al[2], wah[ks], c2); c3 = mfma_bf(al[3], wah[ks], c3);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { ah[q] = nh[q]; al[q] = nl[q]; }
            }
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) m = max3f(m, max3f(c0[r], c1[r], c2[r]), c3[r]);
            float mlo, mhi;
            half_pair(m, mlo, mhi);   // v_permlane32_swap: the LDS is this kernel's scarcest resource, no ds_bpermute
            if constexpr (!ARG) {
                if (h == 0) rm[cb * 32 + j] = fmaxf(rmc, fmaxf(mlo, mhi));
            } else {
                const float mw = fmaxf(mlo, mhi);
                if (__ballot(mw > rmc)) {     // some channel of this block has a new maximum: find its first row
                    int am = 0;               // descending rows, so that the smallest row holding the maximum wins
#pragma unroll
                    for (int r = 15; r >= 0; --r) am = (c3[r] == m) ? 96 + mfma_row(r, lane) : am;
#pragma unroll
                    for (int r = 15; r >= 0; --r) am = (c2[r] == m) ? 64 + mfma_row(r, lane) : am;
#pragma unroll
                    for (int r = 15; r >= 0; --r) am = (c1[r] == m) ? 32 + mfma_row(r, lane) : am;
#pragma unroll
                    for (int r = 15; r >= 0; --r) am = (c0[r] == m) ? mfma_row(r, lane) : am;
                    int alo, ahi;
                    half_pair(am, alo, ahi);
                    const bool hi_wins = mhi > mlo || (mhi == mlo && ahi < alo);
                    const int aw = hi_wins ? ahi : alo;
                    if (h == 0 && mw > rmc) {
                        rm[cb * 32 + j] = mw;
                        int n = tile * XP + aw;
                        ri[cb * 32 + j] = n < N ? n : N - 1;
                    }
                }
            }
        };
        if constexpr (WRES) {
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) block4(wave + 8 * ci, wres[ci], wres[ci]);
        } else {
#pragma unroll 1
            for (int ci = 0; ci < 4; ++ci) {
                load_wx<8, NT>(wah, wal, w3x, wave + 8 * ci, lane);
                block4(wave + 8 * ci, wah, wal);
            }
        }
        TM(7)
        // the barrier after the next tile's layer 1 orders the h2 rewrite after these reads
    }
    TM_END_TO(pngpd_tm_x3)
    if (h == 0) {
        float *o = out + ((size_t)b * S + s) * 1024;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            const int c = (wave + 8 * ci) * 32 + j;
            float v = rm[c] + b3[c];
            if (relu_last) v = fmaxf(v, 0.f);
            o[c] = s_bad ? __builtin_nanf("") : v;
            if (ARG) out_arg[((size_t)b * S + s) * 1024 + c] = ri[c];
        }
    }
}


// ---------------------------------------------------------------------------------------
// OPT-IN bf16x3 variant of the training main pass (pass C, see pngpd_train.hip): same outputs
// (pmax/parg (blk,1024), psum (blk,2,1024)), per-channel affine forms instead of folded weights:
//   h1 = relu((W1 x' + b1)*s1c + t1c),  h2 = relu((W2 h1)*s2c + t2c),  z3s = (sgn*W3) h2
// w2x / w3x = split_pack_bf16 of the RAW (128,64) / sign-folded (1024,128) weights.
// ---------------------------------------------------------------------------------------
#define X3T_LDS_BYTES (X3_LDS_BYTES + 3 * 1024 * 4)
#define X3TZ_LDS_BYTES (2 * XP * X2S * 2 + 4 * 1024 * 4)     // LOADZ: no h1 tiles, no staged points

// LOADZ: z2 = W2 h1 was stored by pass B (z2t, the lane-major 64-point tiles of pngpd_train.hip) and is read back —
// no points, no layer 1 (and none of its 96 wave-uniform constants, which used to cost 195 spilled SGPRs), no layer-2
// MFMAs, no h1 tiles, two barriers per tile instead of three.  A 128-point tile here is two consecutive 64-point tiles
// there, and the lane ownership is the same: thread (sub-tile tid >> 8, t = tid & 255) of this kernel holds exactly
// the 32 values thread t of pass B wrote.  T64 = ceil(N / 64): when it is odd the last tile's second half does not
// exist in z2t; the first half is read twice (duplicates of real points: masked out of the sums by n < N, never a
// new maximum, and a tie is resolved towards the smaller row).
template <int NT, bool LOADZ>
__global__ __launch_bounds__(512, 2) void trunk_fwd_train_x3_kernel(
    const float *__restrict__ x, int N, const float *__restrict__ trans,
    const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ s1c,
    const float *__restrict__ t1c, const u16 *__restrict__ w2x, const float *__restrict__ s2c,
    const float *__restrict__ t2c, const u16 *__restrict__ w3x, int T, int S,
    float *__restrict__ pmax, int *__restrict__ parg, float *__restrict__ psum, float *__restrict__ psh,
    const f32x4 *__restrict__ z2t, int T64) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16 *h1h = (u16 *)smem_raw;
    u16 *h1l = h1h + XP * X1S;
    u16 *h2h = LOADZ ? (u16 *)smem_raw : h1l + XP * X1S;
    u16 *h2l = h2h + XP * X2S;
    float *xs = (float *)(h2l + XP * X2S);
    float *rm = LOADZ ? xs : xs + 3 * XP;
    double hsum = 0.0;   // sum of h2[.][(wave&3)*32 + j] over this wave's valid rows (fp64 across tiles)
    int *ri = (int *)(rm + 1024);
    float *ss = (float *)(ri + 1024);
    float *sq = ss + 1024;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / S, s = blockIdx.x - b * S;
    const int t0 = (int)(((long)s * T) / S), t1 = (int)(((long)(s + 1) * T) / S);
    const float *xb = x + (size_t)b * 3 * N;
    float tm[9] = {0};
    const bool has_t = trans != nullptr;
    if (has_t) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tm[i] = trans[(size_t)b * 9 + i];
    }
    for (int i = tid; i < 1024; i += 512) { rm[i] = -INFINITY; ri[i] = 0; ss[i] = 0.f; sq[i] = 0.f; }
    f32x4 wah[8], wal[8];
    f32x4 zq[8];
    uint4 zb[4];          // NT == 1: pass B stored bf16 tiles (pngpd_bf.h), four quads per lane instead of eight
    auto fetch_z = [&](int tile) {
        int t64 = 2 * tile + (tid >> 8);
        t64 = t64 < T64 ? t64 : 2 * tile;
        if (NT == 1) {
            const uint4 *zt = (const uint4 *)z2t + ((size_t)(b * T64 + t64) * 4) * 256 + (tid & 255);
#pragma unroll
            for (int i = 0; i < 4; ++i) zb[i] = zt[(size_t)i * 256];
        } else {
            const f32x4 *zt = z2t + ((size_t)(b * T64 + t64) * 8) * 256 + (tid & 255);
#pragma unroll
            for (int i = 0; i < 8; ++i) zq[i] = zt[(size_t)i * 256];
        }
    };
    if (LOADZ) fetch_z(t0);
    TM_DECL
    for (int tile = t0; tile < t1; ++tile) {
        const int nbase = tile * XP;
        TM(0)
        if (LOADZ) {
            if (tile > t0) __syncthreads();   // every wave is done reading the previous tile's h2
            TM(1)
        } else {
        if (tid < XP) {
            int n = nbase + tid; n = n < N ? n : N - 1;
            float x0 = xb[n], x1 = xb[N + n], x2 = xb[2 * N + n];
            if (has_t) {
                const float y0 = fmaf(x2, tm[6], fmaf(x1, tm[3], x0 * tm[0]));
                const float y1 = fmaf(x2, tm[7], fmaf(x1, tm[4], x0 * tm[1]));
                const float y2 = fmaf(x2, tm[8], fmaf(x1, tm[5], x0 * tm[2]));
                x0 = y0; x1 = y1; x2 = y2;
            }
            xs[tid] = x0; xs[XP + tid] = x1; xs[2 * XP + tid] = x2;
        }
        __syncthreads();
        {
            const int p = tid & 127, g = wave >> 1;
            const float x0 = xs[p], x1 = xs[XP + p], x2 = xs[2 * XP + p];
            float zv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = g * 16 + e;
                const float z = fmaf(w1[c * 3 + 2], x2, fmaf(w1[c * 3 + 1], x1, fmaf(w1[c * 3], x0, b1[c])));
                zv[e] = fmaxf(fmaf(z, s1c[c], t1c[c]), 0.f);
            }
            uint4 *dh = (uint4 *)(h1h + p * X1S + g * 16), *dl = (uint4 *)(h1l + p * X1S + g * 16);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                // channel pairs through ONE v_cvt_pk_bf16_f32 each (pngpd_bf.h bf_pk2 / bf_split_pk2): same bits as the
                // per-value split2 this replaced, a third of its instructions
                uint4 vh, vl = {0u, 0u, 0u, 0u};
                if (NT == 3) {
                    bf_split_pk2(zv[q * 8 + 0], zv[q * 8 + 1], vh.x, vl.x); bf_split_pk2(zv[q * 8 + 2], zv[q * 8 + 3], vh.y, vl.y);
                    bf_split_pk2(zv[q * 8 + 4], zv[q * 8 + 5], vh.z, vl.z); bf_split_pk2(zv[q * 8 + 6], zv[q * 8 + 7], vh.w, vl.w);
                } else {
                    vh.x = bf_pk2(zv[q * 8 + 0], zv[q * 8 + 1]); vh.y = bf_pk2(zv[q * 8 + 2], zv[q * 8 + 3]);
                    vh.z = bf_pk2(zv[q * 8 + 4], zv[q * 8 + 5]); vh.w = bf_pk2(zv[q * 8 + 6], zv[q * 8 + 7]);
                }
                dh[q] = vh;
                if (NT == 3) dl[q] = vl;
            }
        }
        __syncthreads();
        }   // !LOADZ
        {
            const int cb = wave & 3, pb0 = (wave >> 2) * 2;
            f32x16 a0 = {0}, a1 = {0};
            if (LOADZ) {
                if (NT == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v[8];
                        bf_tile_unpack(zb[i], v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { if (i < 2) a0[8 * i + e] = v[e]; else a1[8 * (i - 2) + e] = v[e]; }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { a0[r] = zq[r >> 2][r & 3]; a1[r] = zq[4 + (r >> 2)][r & 3]; }
                }
                if (tile + 1 < t1) fetch_z(tile + 1);   // in flight during this tile's layer 3
            } else {
            f32x4 w2h[4], w2l[4];
            load_wx<4, NT>(w2h, w2l, w2x, cb, lane);
            const int r0 = (pb0 * 32 + j) * X1S + h * 8, r1 = ((pb0 + 1) * 32 + j) * X1S + h * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 ah0 = *(const f32x4 *)(h1h + r0 + ks * 16), ah1 = *(const f32x4 *)(h1h + r1 + ks * 16);
                a0 = mfma_bf(ah0, w2h[ks], a0); a1 = mfma_bf(ah1, w2h[ks], a1);
                if (NT == 3) {
                    const f32x4 al0 = *(const f32x4 *)(h1l + r0 + ks * 16), al1 = *(const f32x4 *)(h1l + r1 + ks * 16);
                    a0 = mfma_bf(ah0, w2l[ks], a0); a1 = mfma_bf(ah1, w2l[ks], a1);
                    a0 = mfma_bf(al0, w2h[ks], a0); a1 = mfma_bf(al1, w2h[ks], a1);
                }
            }
            }   // !LOADZ
            const float sc = s2c[cb * 32 + j], sh = t2c[cb * 32 + j];
            f32x2 hs2 = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma_row(r, lane);
                unsigned hi, lo = 0u;
                const float v0 = fmaxf(fmaf(a0[r], sc, sh), 0.f), v1 = fmaxf(fmaf(a1[r], sc, sh), 0.f);
                if (NT == 3) bf_split_pk2(v0, v1, hi, lo); else hi = bf_pk2(v0, v1);
                h2h[(pb0 * 32 + row) * X2S + cb * 32 + j] = (u16)hi;
                h2h[((pb0 + 1) * 32 + row) * X2S + cb * 32 + j] = (u16)(hi >> 16);
                if (NT == 3) {
                    h2l[(pb0 * 32 + row) * X2S + cb * 32 + j] = (u16)lo;
                    h2l[((pb0 + 1) * 32 + row) * X2S + cb * 32 + j] = (u16)(lo >> 16);
                }
                if (nbase + XP <= N) {
                    hs2 += f32x2{v0, v1};
                } else {
                    hs2[0] += (nbase + pb0 * 32 + row < N) ? v0 : 0.f;
                    hs2[1] += (nbase + (pb0 + 1) * 32 + row < N) ? v1 : 0.f;
                }
            }
            // finished here: left alone, the compiler sinks these sums below the tile's layer-3 blocks and keeps the 32
            // activations in registers until then
            hsum += (double)(hs2[0] + hs2[1]);
            asm volatile("" : "+v"(hsum));
        }
        TM(2)
        __syncthreads();
        TM(3)
        const bool full = nbase + XP <= N;
#pragma unroll 1
        for (int ci = 0; ci < 4; ++ci) {
            const int cb = wave + 8 * ci;
            load_wx<8, NT>(wah, wal, w3x, cb, lane);
#ifdef PNGPD_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TM(4)
#endif
            float m = -INFINITY, su = 0.f, qu = 0.f;
            int am = 0;
            // all four point blocks at once: 4 independent accumulator chains, A fragments of k-step ks+1 in flight
            // while k-step ks issues (same schedule as trunk_infer_x3_kernel)
            f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
            {
                const int ro = j * X2S + h * 8;
                f32x4 ah[4], al[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ah[q] = *(const f32x4 *)(h2h + ro + q * 32 * X2S);
                    al[q] = (NT == 3) ? *(const f32x4 *)(h2l + ro + q * 32 * X2S) : ah[q];
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    f32x4 nh[4], nl[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        nh[q] = ah[q]; nl[q] = al[q];
                        if (ks < 7) {
                            nh[q] = *(const f32x4 *)(h2h + ro + q * 32 * X2S + (ks + 1) * 16);
                            if (NT == 3) nl[q] = *(const f32x4 *)(h2l + ro + q * 32 * X2S + (ks + 1) * 16);
                        }
                    }
                    c0 = mfma_bf(ah[0], wah[ks], c0); c1 = mfma_bf(ah[1], wah[ks], c1);
                    c2 = mfma_bf(ah[2], wah[ks], c2); c3 = mfma_bf(ah[3], wah[ks], c3);
                    if (NT == 3) {
                        c0 = mfma_bf(ah[0], wal[ks], c0); c1 = mfma_bf(ah[1], wal[ks], c1);
                        c2 = mfma_bf(ah[2], wal[ks], c2); c3 = mfma_bf(ah[3], wal[ks], c3);
                        c0 = mfma_bf(al[0], wah[ks], c0); c1 = mfma_bf(al[1], wah[ks], c1);
                        c2 = mfma_bf(al[2], wah[ks], c2); c3 = mfma_bf(al[3], wah[ks], c3);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) { ah[q] = nh[q]; al[q] = nl[q]; }
                }
            }
            TM(5)
            if (full) {
                // lane_max_moments (pngpd_tile.h): exact first maximum + both moments of 32 values without compares;
                // the two halves of the 128-point tile are merged with "earlier rows win ties"
                float m1, s1, q1; int r1;
                lane_max_moments(c0, c1, m, am, su, qu);
                lane_max_moments(c2, c3, m1, r1, s1, q1);
                if (m1 > m) { m = m1; am = 64 + r1; }
                am += 4 * h;
                su += s1; qu += q1;
            } else {
                auto epi = [&](const f32x16 &ca, const f32x16 &cc, int qp) {
                    // ascending local-row order with strict >: the first maximum wins
#pragma unroll
                    for (int r = 0; r < 16; ++r) { if (ca[r] > m) { m = ca[r]; am = (2 * qp) * 32 + mfma_row(r, lane); } }
#pragma unroll
                    for (int r = 0; r < 16; ++r) { if (cc[r] > m) { m = cc[r]; am = (2 * qp + 1) * 32 + mfma_row(r, lane); } }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = mfma_row(r, lane);
                        const float v0 = (nbase + (2 * qp) * 32 + row < N) ? ca[r] : 0.f;
                        const float v1 = (nbase + (2 * qp + 1) * 32 + row < N) ? cc[r] : 0.f;
                        su += v0 + v1; qu = fmaf(v0, v0, fmaf(v1, v1, qu));
                    }
                };
                epi(c0, c1, 0);
                epi(c2, c3, 1);
            }
            {
                float mlo, mhi; int alo, ahi;
                half_pair(m, mlo, mhi); half_pair(am, alo, ahi);
                const bool hi_wins = mhi > mlo || (mhi == mlo && ahi < alo);
                m = hi_wins ? mhi : mlo; am = hi_wins ? ahi : alo;
                su = half_sum(su); qu = half_sum(qu);
            }
            if (h == 0) {
                const int c = cb * 32 + j;
                if (m > rm[c]) { rm[c] = m; const int n = nbase + am; ri[c] = n < N ? n : N - 1; }
                ss[c] += su; sq[c] += qu;
            }
            TM(6)
        }
    }
    TM_END_TO(pngpd_tm_x3)
    hsum += __shfl_xor(hsum, 32);
    if (h == 0) {
        psh[((size_t)blockIdx.x * 2 + (wave >> 2)) * 128 + (wave & 3) * 32 + j] = (float)hsum;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            const int c = (wave + 8 * ci) * 32 + j;
            pmax[(size_t)blockIdx.x * 1024 + c] = rm[c];
            parg[(size_t)blockIdx.x * 1024 + c] = ri[c];
            psum[((size_t)blockIdx.x * 2) * 1024 + c] = ss[c];
            psum[((size_t)blockIdx.x * 2 + 1) * 1024 + c] = sq[c];
        }
    }
}

__global__ void pool_reduce_x3_kernel(const float *__restrict__ part, int S, float *__restrict__ out, int total) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int b = idx >> 10, c = idx & 1023;
    const float *p = part + (size_t)b * S * 1024 + c;
    float m = p[0];
    bool nan = m != m;
    for (int s = 1; s < S; ++s) { const float v = p[(size_t)s * 1024]; nan |= v != v; m = fmaxf(m, v); }
    out[idx] = nan ? __builtin_nanf("") : m;
}

// the same with the arg-max carried along (the earliest split wins ties; a NaN partial poisons the row as above)
__global__ void pool_reduce_arg_x3_kernel(const float *__restrict__ part, const int *__restrict__ parg, int S,
                                          float *__restrict__ out, int *__restrict__ out_arg, int total) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int b = idx >> 10, c = idx & 1023;
    const float *p = part + (size_t)b * S * 1024 + c;
    const int *a = parg + (size_t)b * S * 1024 + c;
    float m = p[0];
    int am = a[0];
    bool nan = m != m;
    for (int s = 1; s < S; ++s) {
        const float v = p[(size_t)s * 1024];
        nan |= v != v;
        if (v > m) { m = v; am = a[(size_t)s * 1024]; }
    }
    out[idx] = nan ? __builtin_nanf("") : m;
    out_arg[idx] = am;
}

#define X3_DEFAULT_TARGET_BLOCKS 1024

template <int NT, bool XBF, bool ARG>
static int launch_infer_bf(const void *x, int B, int N, const float *trans, const float *w1, const float *b1,
                           const u16 *w2x, const float *b2, const u16 *w3x, const float *b3, int relu_last, int T,
                           int S, float *dst, int *dst_arg, hipStream_t stream) {
    const size_t lds = X3_LDS_BYTES + (ARG ? 4096 : 0);
    int st = pngpd_allow_lds((const void *)trunk_infer_x3_kernel<NT, XBF, ARG>, lds);
    if (st != PNGPD_OK) return st;
    hipLaunchKernelGGL((trunk_infer_x3_kernel<NT, XBF, ARG>), dim3((unsigned)B * S), dim3(512), lds, stream,
                       x, N, trans, w1, b1, w2x, b2, w3x, b3, relu_last, T, S, dst, dst_arg);
    return pngpd_launch_status();
}

template <int NT, bool LOADZ>
static int launch_train_bf(const float *x, int B, int N, const float *trans, const float *w1, const float *b1,
                           const float *s1c, const float *t1c, const u16 *w2x, const float *s2c, const float *t2c,
                           const u16 *w3sx, int T, int S, float *pmax, int *parg, float *psum, float *psh,
                           const float *z2t, hipStream_t stream) {
    const size_t lds = LOADZ ? X3TZ_LDS_BYTES : X3T_LDS_BYTES;
    int st = pngpd_allow_lds((const void *)trunk_fwd_train_x3_kernel<NT, LOADZ>, lds);
    if (st != PNGPD_OK) return st;
    hipLaunchKernelGGL((trunk_fwd_train_x3_kernel<NT, LOADZ>), dim3((unsigned)B * S), dim3(512), lds, stream,
                       x, N, trans, w1, b1, s1c, t1c, w2x, s2c, t2c, w3sx, T, S, pmax, parg, psum, psh,
                       (const f32x4 *)z2t, (N + 63) / 64);
    return pngpd_launch_status();
}

extern "C" {

int pngpd_trunk_infer_bf_splits(int B, int N, int target_blocks) {
    if (B <= 0 || N <= 0) return 0;
    return pngpd_splits_for(B, (N + XP - 1) / XP, target_blocks > 0 ? target_blocks : X3_DEFAULT_TARGET_BLOCKS);
}

int pngpd_split_pack_bf16(const float *W, int C, int K, void *out, void *stream) {
    if (!W || !out || C <= 0 || K <= 0 || (C & 31) || (K & 15)) return PNGPD_ERR_INVALID_ARG;
    const int total = C * K;
    hipLaunchKernelGGL(split_pack_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       W, C, K, (u16 *)out);
    return pngpd_launch_status();
}

static int infer_bf_impl(const void *x, int x_is_bf16, int B, int N, const float *trans,
                         const float *w1, const float *b1, const void *w2x, const float *b2,
                         const void *w3x, const float *b3, int relu_last, int nterms, int splits,
                         float *out_pool, int *out_arg, void *workspace, size_t workspace_bytes, void *stream) {
    if (!x || !w1 || !b1 || !w2x || !b2 || !w3x || !b3 || !out_pool || B <= 0 || N <= 0 ||
        (nterms != 1 && nterms != 3))
        return PNGPD_ERR_INVALID_ARG;
    const int T = (N + XP - 1) / XP;
    const int S = (splits > 0) ? (splits > T ? T : splits) : pngpd_splits_for(B, T, X3_DEFAULT_TARGET_BLOCKS);
    float *dst = out_pool;
    int *dst_arg = out_arg;
    if (S > 1) {
        const size_t need = (size_t)B * S * 1024 * (sizeof(float) + (out_arg ? sizeof(int) : 0));
        if (!workspace || workspace_bytes < need) return PNGPD_ERR_WORKSPACE;
        dst = (float *)workspace;
        if (out_arg) dst_arg = (int *)(dst + (size_t)B * S * 1024);
    }
    const u16 *w2 = (const u16 *)w2x, *w3 = (const u16 *)w3x;
    hipStream_t sm = (hipStream_t)stream;
    int st;
#define PNGPD_X3_LAUNCH(NT_, XBF_, ARG_) \
    launch_infer_bf<NT_, XBF_, ARG_>(x, B, N, trans, w1, b1, w2, b2, w3, b3, relu_last, T, S, dst, dst_arg, sm)
    if (out_arg) {
        if (nterms == 3) st = x_is_bf16 ? PNGPD_X3_LAUNCH(3, true, true) : PNGPD_X3_LAUNCH(3, false, true);
        else st = x_is_bf16 ? PNGPD_X3_LAUNCH(1, true, true) : PNGPD_X3_LAUNCH(1, false, true);
    } else {
        if (nterms == 3) st = x_is_bf16 ? PNGPD_X3_LAUNCH(3, true, false) : PNGPD_X3_LAUNCH(3, false, false);
        else st = x_is_bf16 ? PNGPD_X3_LAUNCH(1, true, false) : PNGPD_X3_LAUNCH(1, false, false);
    }
#undef PNGPD_X3_LAUNCH
    if (st != PNGPD_OK) return st;
    if (S > 1) {
        const int total = B * 1024;
        if (out_arg)
            hipLaunchKernelGGL(pool_reduce_arg_x3_kernel, dim3((total + 255) / 256), dim3(256), 0, sm,
                               (const float *)dst, (const int *)dst_arg, S, out_pool, out_arg, total);
        else
            hipLaunchKernelGGL(pool_reduce_x3_kernel, dim3((total + 255) / 256), dim3(256), 0, sm,
                               (const float *)workspace, S, out_pool, total);
        st = pngpd_launch_status();
    }
    return st;
}

int pngpd_trunk_fwd_infer_bf(const void *x, int x_is_bf16, int B, int N, const float *trans,
                             const float *w1, const float *b1, const void *w2x, const float *b2,
                             const void *w3x, const float *b3, int relu_last, int nterms, int splits,
                             float *out_pool, void *workspace, size_t workspace_bytes, void *stream) {
    return infer_bf_impl(x, x_is_bf16, B, N, trans, w1, b1, w2x, b2, w3x, b3, relu_last, nterms, splits, out_pool,
                         nullptr, workspace, workspace_bytes, stream);
}

int pngpd_trunk_fwd_infer_bf_arg(const void *x, int x_is_bf16, int B, int N, const float *trans,
                                 const float *w1, const float *b1, const void *w2x, const float *b2,
                                 const void *w3x, const float *b3, int relu_last, int nterms, int splits,
                                 float *out_pool, int *out_arg, void *workspace, size_t workspace_bytes,
                                 void *stream) {
    if (!out_arg) return PNGPD_ERR_INVALID_ARG;
    return infer_bf_impl(x, x_is_bf16, B, N, trans, w1, b1, w2x, b2, w3x, b3, relu_last, nterms, splits, out_pool,
                         out_arg, workspace, workspace_bytes, stream);
}

int pngpd_trunk_fwd_train_bf(const float *x, int B, int N, const float *trans,
                             const float *w1, const float *b1, const float *s1c, const float *t1c,
                             const void *w2x, const float *s2c, const float *t2c, const void *w3sx, int nterms, int S,
                             float *pmax, int *parg, float *psum, float *psh, const void *z2tv, void *stream) {
    const float *z2t = (const float *)z2tv;
    if (!x || !w1 || !b1 || !s1c || !t1c || !w2x || !s2c || !t2c || !w3sx || !pmax || !parg || !psum || !psh ||
        B <= 0 || N <= 0 || (nterms != 1 && nterms != 3))
        return PNGPD_ERR_INVALID_ARG;
    const int T = (N + XP - 1) / XP;
    if (S < 1 || S > T) return PNGPD_ERR_INVALID_ARG;   // S: the split count the caller sized pmax/parg/psum for
    const u16 *w2 = (const u16 *)w2x, *w3 = (const u16 *)w3sx;
    hipStream_t sm = (hipStream_t)stream;
    if (z2t)
        return nterms == 3
            ? launch_train_bf<3, true>(x, B, N, trans, w1, b1, s1c, t1c, w2, s2c, t2c, w3, T, S, pmax, parg, psum, psh, z2t, sm)
            : launch_train_bf<1, true>(x, B, N, trans, w1, b1, s1c, t1c, w2, s2c, t2c, w3, T, S, pmax, parg, psum, psh, z2t, sm);
    return nterms == 3
        ? launch_train_bf<3, false>(x, B, N, trans, w1, b1, s1c, t1c, w2, s2c, t2c, w3, T, S, pmax, parg, psum, psh, nullptr, sm)
        : launch_train_bf<1, false>(x, B, N, trans, w1, b1, s1c, t1c, w2, s2c, t2c, w3, T, S, pmax, parg, psum, psh, nullptr, sm);
}

}  // extern "C"
