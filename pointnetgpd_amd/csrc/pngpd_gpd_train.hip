// libpngpd — backward of the GPD comparator's convolution stages:
//   GPDClassifier  conv(5x5) -> MaxPool2d(2,2), twice      PointNetGPD/model/gpd.py:13-24
//   trained by     loss.backward()                         PointNetGPD/main_1v_gpd.py:105
// The forward (pngpd_conv5_pool2_arg, pngpd_gpd.hip) records which of the four window positions each pooled pixel took;
// the gradient of a stage is therefore SPARSE in the convolution's output — one position per pooled pixel — and no kernel
// below materialises the dense form in HBM:
//   * weights / bias: the pooled gradients become a list of (plane offset, value) pairs (8 bytes per pooled pixel, the only
//     intermediate); a workgroup owns (5 output channels, <= 5 input planes, a slice of the batch), stages the planes in
//     LDS, a lane owns one or two (plane, ky, kx) taps and its wave walks a quarter of each channel's list, read through
//     the scalar unit.  Per-slice partial sums, reduced in a fixed order by a second launch: deterministic, no atomics.
//   * input (the second stage only — the first stage's input is the image): a workgroup owns (sample, 4 input planes);
//     the sparse gradient of 25 output channels is expanded into zero-padded planes in LDS and every thread evaluates
//     the full correlation for a 1x4 strip of pixels; the channel's 100 weights arrive through the scalar unit.
// Small, latency-bound VALU kernels: the comparator is not the hot path, these exist so that a CUDA tensor
// in train() mode has a libpngpd path instead of ATen / MIOpen.
#include "pngpd_common.h"

#define C5_OCG 5
#define C5_CCH 5
#define C5_ICG 4

// the sparse gradient of a stage as a list: per pooled pixel (plane offset of the window position it came from, value)
__global__ __launch_bounds__(256) void conv5_pool_list_kernel(const float *__restrict__ dout,
                                                              const unsigned char *__restrict__ arg, int Hin, long long n,
                                                              int2 *__restrict__ lst) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int Hp = (Hin - 4) / 2, pp = (int)(i % (Hp * Hp)), code = arg[i], py = pp / Hp, px = pp - py * Hp;
    lst[i] = make_int2((2 * py + (code >> 1)) * Hin + 2 * px + (code & 1), __float_as_int(dout[i]));
}

// Weights / bias.  The list entries are the same for every lane of a wave (a wave owns a contiguous quarter of a channel's
// pooled pixels, its lanes own the taps), so they arrive through the SCALAR unit — s_load from the list in global memory —
// and the only LDS access per multiply-add is the gather from the staged plane at (tap base + entry offset).
template <int OCG, int CCH>
__global__ __launch_bounds__(256) void conv5_pool2_bwd_w_kernel(
    const float *__restrict__ in, int Cin, int Hin, const int2 *__restrict__ lst, int Cout, int B, int S,
    float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int TJ = (CCH * 25 + 63) / 64;           // taps per lane
    constexpr int RED = 4 * TJ * 64 * OCG + 4 * OCG;   // cross-wave reduction buffers (alias the planes at the end)
    const int Hp = (Hin - 4) / 2, HP2 = Hp * Hp, HH = Hin * Hin;
    float *plane = sm;                                 // [CCH][HH]
    const int nchunks = (Cin + CCH - 1) / CCH;
    const int og = blockIdx.x / nchunks, ch = blockIdx.x - og * nchunks;
    const int s = blockIdx.y;
    const int c0 = ch * CCH, nc = (Cin - c0) < CCH ? (Cin - c0) : CCH;
    const int NJ = nc * 25;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int jbase[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
        const int j = lane + 64 * t;
        const int jc = j / 25, k = j - jc * 25, ky = k / 5, kx = k - ky * 5;
        jbase[t] = j < NJ ? jc * HH + ky * Hin + kx : 0;   // idle lanes read a valid address; their sums are dropped
    }
    float acc[TJ][OCG], dbq[OCG];
#pragma unroll
    for (int q = 0; q < OCG; ++q) {
        dbq[q] = 0.f;
#pragma unroll
        for (int t = 0; t < TJ; ++t) acc[t][q] = 0.f;
    }
    const int per_wave = (HP2 + 3) / 4;
    const int pp0 = wave * per_wave, pp1 = (pp0 + per_wave) < HP2 ? (pp0 + per_wave) : HP2;
    const int b0 = (int)((long long)B * s / S), b1 = (int)((long long)B * (s + 1) / S);
    const size_t pstride = (size_t)Cout * Cin * 25 + Cout;
    for (int b = b0; b < b1; ++b) {
        __syncthreads();
        const float *inb = in + ((size_t)b * Cin + c0) * HH;
        for (int i = tid; i < nc * HH; i += 256) plane[i] = inb[i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < OCG; ++q) {
            const int oc = og * OCG + q;
            if (oc < Cout) {
                const int2 *L = lst + ((size_t)b * Cout + oc) * HP2;
                float dbs = 0.f;
                // (issuing the next entries' scalar loads ahead of the gathers by hand — groups of 4, double-buffered —
                //  measured 40-60 % slower than leaving the schedule to the compiler)
#pragma unroll 4
                for (int pp = pp0; pp < pp1; ++pp) {
                    const int2 e = L[pp];
                    const float g = __int_as_float(e.y);
#pragma unroll
                    for (int t = 0; t < TJ; ++t) acc[t][q] = fmaf(g, plane[jbase[t] + e.x], acc[t][q]);
                    dbs += g;
                }
                dbq[q] += dbs;
            }
        }
    }
    __syncthreads();
    float *red = plane, *dbred = plane + 4 * TJ * 64 * OCG;
#pragma unroll
    for (int t = 0; t < TJ; ++t)
#pragma unroll
        for (int q = 0; q < OCG; ++q) red[((wave * TJ + t) * 64 + lane) * OCG + q] = acc[t][q];
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < OCG; ++q) dbred[wave * OCG + q] = dbq[q];
    __syncthreads();
    float *ps_out = part + (size_t)s * pstride;
    if (tid < NJ) {
        const int t = tid >> 6, l = tid & 63, jc = tid / 25, k = tid - jc * 25;
#pragma unroll
        for (int q = 0; q < OCG; ++q) {
            float v = 0.f;
            for (int w = 0; w < 4; ++w) v += red[((w * TJ + t) * 64 + l) * OCG + q];
            const int oc = og * OCG + q;
            if (oc < Cout) ps_out[((size_t)oc * Cin + c0 + jc) * 25 + k] = v;
        }
    }
    if (ch == 0 && tid < OCG) {
        const int oc = og * OCG + tid;
        if (oc < Cout) ps_out[(size_t)Cout * Cin * 25 + oc] = dbred[tid] + dbred[OCG + tid] + dbred[2 * OCG + tid] + dbred[3 * OCG + tid];
    }
    (void)RED;
}

// Few taps (the first stage on 3-channel images: 75 taps per output channel): lanes cannot all own a tap of ONE list
// position, so this form keeps the (offset, value) lists of 5 channels in LDS and spreads the threads over (tap, list
// slice) instead; built from `arg` / `dout` directly.
template <int OCG, int CCH>
__global__ __launch_bounds__(256) void conv5_pool2_bwd_w_lds_kernel(
    const float *__restrict__ in, int Cin, int Hin, const float *__restrict__ dout,
    const unsigned char *__restrict__ arg, int Cout, int B, int S, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Hp = (Hin - 4) / 2, HP2 = Hp * Hp, HH = Hin * Hin;
    const int plane_floats = CCH * HH > 256 * OCG ? CCH * HH : 256 * OCG;
    float *plane = sm;                                 // [CCH][HH]; at the end the cross-slice reduction buffer
    int2 *lst = (int2 *)(plane + plane_floats);        // [OCG][HP2] (offset of the chosen window position in a plane,
                                                       //  pooled gradient): one 8-byte LDS read per list entry
    const int nchunks = (Cin + CCH - 1) / CCH;
    const int og = blockIdx.x / nchunks, ch = blockIdx.x - og * nchunks;
    const int s = blockIdx.y;
    const int c0 = ch * CCH, nc = (Cin - c0) < CCH ? (Cin - c0) : CCH;
    const int NJ = nc * 25, PS = 256 / NJ;             // taps of this workgroup; slices of the list walked in parallel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = tid % NJ, ps = tid / NJ;
    const bool act = ps < PS;
    const int jc = j / 25, k = j - jc * 25, ky = k / 5, kx = k - ky * 5;
    const int jbase = jc * HH + ky * Hin + kx;
    float acc[OCG];
#pragma unroll
    for (int q = 0; q < OCG; ++q) acc[q] = 0.f;
    float dbq[(OCG + 3) / 4];
#pragma unroll
    for (int q = 0; q < (OCG + 3) / 4; ++q) dbq[q] = 0.f;
    const int b0 = (int)((long long)B * s / S), b1 = (int)((long long)B * (s + 1) / S);
    const size_t pstride = (size_t)Cout * Cin * 25 + Cout;
    for (int b = b0; b < b1; ++b) {
        __syncthreads();
        const float *inb = in + ((size_t)b * Cin + c0) * HH;
        for (int i = tid; i < nc * HH; i += 256) plane[i] = inb[i];
        for (int i = tid; i < OCG * HP2; i += 256) {
            const int q = i / HP2, pp = i - q * HP2, oc = og * OCG + q;
            if (oc < Cout) {
                const size_t idx = ((size_t)b * Cout + oc) * HP2 + pp;
                const int code = arg[idx], py = pp / Hp, px = pp - py * Hp;
                lst[i] = make_int2((2 * py + (code >> 1)) * Hin + 2 * px + (code & 1), __float_as_int(dout[idx]));
            } else {
                lst[i] = make_int2(0, 0);
            }
        }
        __syncthreads();
        if (act) {
#pragma unroll 4
            for (int pp = ps; pp < HP2; pp += PS) {
#pragma unroll
                for (int q = 0; q < OCG; ++q) {
                    const int2 e = lst[q * HP2 + pp];
                    acc[q] = fmaf(__int_as_float(e.y), plane[jbase + e.x], acc[q]);
                }
            }
        }
        if (ch == 0) {                                 // bias: wave w sums the lists of channels w, w + 4, ...
#pragma unroll
            for (int r = 0; r < (OCG + 3) / 4; ++r) {
                const int q = wave + 4 * r;
                if (q < OCG) {
                    float t = 0.f;
                    for (int pp = lane; pp < HP2; pp += 64) t += __int_as_float(lst[q * HP2 + pp].y);
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m);
                    dbq[r] += t;
                }
            }
        }
    }
    __syncthreads();
    if (act) {
#pragma unroll
        for (int q = 0; q < OCG; ++q) plane[(ps * NJ + j) * OCG + q] = acc[q];
    }
    __syncthreads();
    float *ps_out = part + (size_t)s * pstride;
    if (tid < NJ) {
#pragma unroll
        for (int q = 0; q < OCG; ++q) {
            float t = 0.f;
            for (int p2 = 0; p2 < PS; ++p2) t += plane[(p2 * NJ + tid) * OCG + q];
            const int oc = og * OCG + q;
            if (oc < Cout) ps_out[((size_t)oc * Cin + c0 + jc) * 25 + k] = t;
        }
    }
    if (ch == 0 && lane == 0) {
#pragma unroll
        for (int r = 0; r < (OCG + 3) / 4; ++r) {
            const int q = wave + 4 * r, oc = og * OCG + q;
            if (q < OCG && oc < Cout) ps_out[(size_t)Cout * Cin * 25 + oc] = dbq[r];
        }
    }
}

// dW (nW) | db (Cout) = sum over the S batch slices, in slice order, accumulated in fp64
__global__ __launch_bounds__(256) void conv5_bwd_reduce_kernel(const float *__restrict__ part, int S, int nW, int Cout,
                                                               float *__restrict__ dW, float *__restrict__ db) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nW + Cout) return;
    double t = 0.0;
    for (int s = 0; s < S; ++s) t += (double)part[(size_t)s * (nW + Cout) + i];
    if (i < nW) dW[i] = (float)t;
    else db[i - nW] = (float)t;
}

template <int ICG>
__global__ __launch_bounds__(256) void conv5_pool2_bwd_x_kernel(
    const float *__restrict__ dout, const unsigned char *__restrict__ arg, const float *__restrict__ W, int Cin, int Hin,
    int Cout, int OCC, float *__restrict__ din) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Hp = (Hin - 4) / 2, HP2 = Hp * Hp, PW = Hin + 4, PP = PW * PW;
    float *pl = sm;                                    // [OCC][PP]: d(conv output) at (y + 4, x + 4), zeros around
    const int nicg = (Cin + ICG - 1) / ICG;
    const int b = blockIdx.x / nicg, ic0 = (blockIdx.x - b * nicg) * ICG;
    const int SW = Hin / 4, tid = threadIdx.x;
    const bool act = tid < Hin * SW;
    const int Y = act ? tid / SW : 0, X0 = act ? (tid - (tid / SW) * SW) * 4 : 0;
    float acc[4][ICG];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int ic = 0; ic < ICG; ++ic) acc[d][ic] = 0.f;
    for (int oc0 = 0; oc0 < Cout; oc0 += OCC) {
        const int noc = (Cout - oc0) < OCC ? (Cout - oc0) : OCC;
        __syncthreads();
        for (int i = tid; i < noc * PP; i += 256) pl[i] = 0.f;
        __syncthreads();
        for (int i = tid; i < noc * HP2; i += 256) {
            const int q = i / HP2, pp = i - q * HP2;
            const size_t idx = ((size_t)b * Cout + oc0 + q) * HP2 + pp;
            const int code = arg[idx], py = pp / Hp, px = pp - py * Hp;
            pl[(size_t)q * PP + (2 * py + (code >> 1) + 4) * PW + 2 * px + (code & 1) + 4] = dout[idx];
        }
        __syncthreads();
        if (act)
            for (int q = 0; q < noc; ++q) {
                const float *P = pl + (size_t)q * PP;
                // the 100 weights of (channel, 4 planes) are the same for every thread: read from global memory at a
                // uniform address they come through the scalar unit and cost no LDS access
                const float *Wq = W + ((size_t)(oc0 + q) * Cin + ic0) * 25;
#pragma unroll
                for (int ky = 0; ky < 5; ++ky) {
                    // din[Y][X0 + d] += dconv[Y - ky][X0 + d - kx] * w[ky][kx]: padded row Y - ky + 4, columns X0 + (d - kx + 4)
                    const float4 r0 = *(const float4 *)(P + (Y - ky + 4) * PW + X0);
                    const float4 r1 = *(const float4 *)(P + (Y - ky + 4) * PW + X0 + 4);
                    const float row[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
                    for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                        for (int ic = 0; ic < ICG; ++ic) {
                            const float w = Wq[(ic0 + ic < Cin ? ic : 0) * 25 + ky * 5 + kx];
#pragma unroll
                            for (int d = 0; d < 4; ++d) acc[d][ic] = fmaf(row[d - kx + 4], w, acc[d][ic]);
                        }
                }
            }
    }
    if (act)
#pragma unroll
        for (int ic = 0; ic < ICG; ++ic)
            if (ic0 + ic < Cin) {
                float4 o = {acc[0][ic], acc[1][ic], acc[2][ic], acc[3][ic]};
                *(float4 *)(din + (((size_t)b * Cin + ic0 + ic) * Hin + Y) * Hin + X0) = o;
            }
}

// backward of F.relu given its OUTPUT: g <- g * (y > 0), in place (gpd.py:27)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float *__restrict__ y, float *__restrict__ g, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) g[i] = y[i] > 0.f ? g[i] : 0.f;
}

// fc1 of the classifier: (B,7200) x (500,7200)^T is 32 output tiles at B = 64, each a 3,600-MFMA chain when one workgroup
// owns a tile (pngpd_fc_fwd: 65 us).  Here K is split over KS workgroups per tile (and their four waves): partial tiles in
// `part` (KS,B,Nout), summed in slice order with the bias / ReLU by a second launch.  Deterministic.
__global__ __launch_bounds__(256) void fc_splitk_kernel(const float *__restrict__ in, int B, int K,
                                                        const float *__restrict__ W, int Nout, int KS,
                                                        float *__restrict__ part) {
    __shared__ float red[3 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int rb = blockIdx.x, cb = blockIdx.y, ks = blockIdx.z;
    int row = rb * 32 + j; row = row < B ? row : B - 1;
    int col = cb * 32 + j; col = col < Nout ? col : Nout - 1;
    const f32x4 *ap = (const f32x4 *)(in + (size_t)row * K) + h;
    const f32x4 *wp = (const f32x4 *)(W + (size_t)col * K) + h;
    const long KBall = K >> 3, U = (long)KS * 4, u = (long)ks * 4 + wave;
    int kb = (int)(KBall * u / U);
    const int KB = (int)(KBall * (u + 1) / U);
    f32x16 acc = {0};
    for (; kb + 4 <= KB; kb += 4) {
        f32x4 a[4], w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] = ap[(kb + q) * 2]; w[q] = wp[(kb + q) * 2]; }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = mfma32(a[q][t], w[q][t], acc);
    }
    for (; kb < KB; ++kb) {
        const f32x4 a = ap[kb * 2], w = wp[kb * 2];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = mfma32(a[t], w[t], acc);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
    const int c = cb * 32 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = acc[r] + red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
        const int orow = rb * 32 + mfma_row(r, lane);
        if (c < Nout && orow < B) part[((size_t)ks * B + orow) * Nout + c] = v;
    }
}

__global__ __launch_bounds__(256) void fc_splitk_finish_kernel(const float *__restrict__ part, int KS, int B, int Nout,
                                                               const float *__restrict__ bias, int relu,
                                                               float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * Nout) return;
    float v = part[i];
    for (int s = 1; s < KS; ++s) v += part[(size_t)s * B * Nout + i];
    v += bias[i % Nout];
    if (relu) v = (v < 0.f) ? 0.f : v;              // NaN-propagating like F.relu
    out[i] = v;
}

static int fc_splitk_slices(int B, int K, int Nout) {
    const long tiles = (long)((B + 31) / 32) * ((Nout + 31) / 32);
    long KS = (512 + tiles - 1) / tiles;            // about two workgroups per CU
    const long most = (K >> 3) / 16;                // at least 4 eight-wide blocks per wave
    if (KS > most) KS = most;
    return KS < 1 ? 1 : (int)KS;
}

static int conv5_bwd_splits(int B, int Cin, int Cout, int cch) {
    const int wgs = ((Cout + C5_OCG - 1) / C5_OCG) * ((Cin + cch - 1) / cch);
    int S = (1024 + wgs - 1) / wgs;
    if (S > B) S = B;
    return S < 1 ? 1 : S;
}

static size_t conv5_bwd_partial_bytes(int B, int Cin, int Cout) {
    const int s2 = conv5_bwd_splits(B, Cin, Cout, 2), s5 = conv5_bwd_splits(B, Cin, Cout, C5_CCH);   // whatever Hin selects
    const size_t bytes = (size_t)(s2 > s5 ? s2 : s5) * ((size_t)Cout * Cin * 25 + Cout) * sizeof(float);
    return (bytes + 255) & ~(size_t)255;
}

extern "C" {

size_t pngpd_conv5_pool2_bwd_workspace_bytes(int B, int Cin, int Hin, int Cout) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Hin < 6) return 0;
    const int Hp = (Hin - 4) / 2;
    return conv5_bwd_partial_bytes(B, Cin, Cout) + (size_t)B * Cout * Hp * Hp * sizeof(int2);   // partials | gradient list
}

int pngpd_conv5_pool2_bwd(const float *in, int B, int Cin, int Hin, const float *W, int Cout, const float *dout,
                          const unsigned char *arg, float *dW, float *db, float *din, void *workspace,
                          size_t workspace_bytes, void *stream) {
    if (!in || !W || !dout || !arg || !dW || !db || !workspace || B <= 0 || Cin <= 0 || Cout <= 0 || Hin < 6 ||
        ((Hin - 4) & 1))
        return PNGPD_ERR_INVALID_ARG;
    if (workspace_bytes < pngpd_conv5_pool2_bwd_workspace_bytes(B, Cin, Hin, Cout)) return PNGPD_ERR_WORKSPACE;
    if (din && ((Hin & 3) || Hin > 32)) return PNGPD_ERR_UNSUPPORTED;   // the strip mapping of the input-gradient kernel
    hipStream_t st = (hipStream_t)stream;
    const int Hp = (Hin - 4) / 2, HP2 = Hp * Hp, HH = Hin * Hin;
    if (Cin * 25 < 100) {                      // few taps: the LDS-list form, two planes per workgroup
        constexpr int cch = 2;
        const int S = conv5_bwd_splits(B, Cin, Cout, cch);
        const int plane_floats = cch * HH > 256 * C5_OCG ? cch * HH : 256 * C5_OCG;
        const size_t lds = ((size_t)plane_floats + 2 * (size_t)C5_OCG * HP2) * sizeof(float);
        if (lds > 150 * 1024) return PNGPD_ERR_UNSUPPORTED;
        int rc = pngpd_allow_lds((const void *)conv5_pool2_bwd_w_lds_kernel<C5_OCG, cch>, lds);
        if (rc != PNGPD_OK) return rc;
        const dim3 grid(((Cout + C5_OCG - 1) / C5_OCG) * ((Cin + cch - 1) / cch), S);
        hipLaunchKernelGGL((conv5_pool2_bwd_w_lds_kernel<C5_OCG, cch>), grid, dim3(256), lds, st, in, Cin, Hin, dout, arg,
                           Cout, B, S, (float *)workspace);
        const int n = Cout * Cin * 25 + Cout;
        hipLaunchKernelGGL(conv5_bwd_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float *)workspace, S,
                           Cout * Cin * 25, Cout, dW, db);
    } else {
        const long long nl = (long long)B * Cout * HP2;
        int2 *lst = (int2 *)((char *)workspace + conv5_bwd_partial_bytes(B, Cin, Cout));
        hipLaunchKernelGGL(conv5_pool_list_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, dout, arg, Hin, nl, lst);
        // big images (the first stage, 60x60): 2 planes per workgroup (29 KB of LDS: five workgroups per CU, one stages
        // while the others compute); small ones (the second stage): 5 planes = 125 taps, two per lane
        const int cch = Hin > 32 ? 2 : C5_CCH;
        const int S = conv5_bwd_splits(B, Cin, Cout, cch);
        const int tj = (cch * 25 + 63) / 64, red = 4 * tj * 64 * C5_OCG + 4 * C5_OCG;
        const size_t lds = (size_t)(cch * HH > red ? cch * HH : red) * sizeof(float);
        if (lds > 150 * 1024) return PNGPD_ERR_UNSUPPORTED;
        const void *fn = cch == 2 ? (const void *)conv5_pool2_bwd_w_kernel<C5_OCG, 2>
                                  : (const void *)conv5_pool2_bwd_w_kernel<C5_OCG, C5_CCH>;
        int rc = pngpd_allow_lds(fn, lds);
        if (rc != PNGPD_OK) return rc;
        const dim3 grid(((Cout + C5_OCG - 1) / C5_OCG) * ((Cin + cch - 1) / cch), S);
        if (cch == 2)
            hipLaunchKernelGGL((conv5_pool2_bwd_w_kernel<C5_OCG, 2>), grid, dim3(256), lds, st, in, Cin, Hin, lst, Cout, B, S,
                               (float *)workspace);
        else
            hipLaunchKernelGGL((conv5_pool2_bwd_w_kernel<C5_OCG, C5_CCH>), grid, dim3(256), lds, st, in, Cin, Hin, lst, Cout,
                               B, S, (float *)workspace);
        const int n = Cout * Cin * 25 + Cout;
        hipLaunchKernelGGL(conv5_bwd_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float *)workspace, S,
                           Cout * Cin * 25, Cout, dW, db);
    }
    if (din) {
        const int PW = Hin + 4;
        const size_t per_oc = (size_t)PW * PW * sizeof(float);
        int OCC = (int)((48 * 1024) / per_oc);      // 10 channels of a 28x28 stage: three workgroups per CU overlap their staging
        if (OCC > Cout) OCC = Cout;
        if (OCC < 1) return PNGPD_ERR_UNSUPPORTED;
        const size_t lds = OCC * per_oc;
        int rc = pngpd_allow_lds((const void *)conv5_pool2_bwd_x_kernel<C5_ICG>, lds);
        if (rc != PNGPD_OK) return rc;
        hipLaunchKernelGGL((conv5_pool2_bwd_x_kernel<C5_ICG>), dim3((unsigned)B * ((Cin + C5_ICG - 1) / C5_ICG)), dim3(256),
                           lds, st, dout, arg, W, Cin, Hin, Cout, OCC, din);
    }
    return pngpd_launch_status();
}

int pngpd_relu_bwd(const float *y, float *g, long long n, void *stream) {
    if (!y || !g || n <= 0) return PNGPD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, g, n);
    return pngpd_launch_status();
}

size_t pngpd_fc_fwd_splitk_workspace_bytes(int B, int K, int Nout) {
    if (B <= 0 || K <= 0 || Nout <= 0) return 0;
    return (size_t)fc_splitk_slices(B, K, Nout) * B * Nout * sizeof(float);
}

int pngpd_fc_fwd_splitk(const float *in, int B, int K, const float *W, const float *bias, int Nout, int relu, float *out,
                        void *workspace, size_t workspace_bytes, void *stream) {
    if (!in || !W || !bias || !out || !workspace || B <= 0 || K <= 0 || Nout <= 0 || (K & 7)) return PNGPD_ERR_INVALID_ARG;
    if (workspace_bytes < pngpd_fc_fwd_splitk_workspace_bytes(B, K, Nout)) return PNGPD_ERR_WORKSPACE;
    const int KS = fc_splitk_slices(B, K, Nout);
    hipLaunchKernelGGL(fc_splitk_kernel, dim3((B + 31) / 32, (Nout + 31) / 32, KS), dim3(256), 0, (hipStream_t)stream, in, B, K,
                       W, Nout, KS, (float *)workspace);
    hipLaunchKernelGGL(fc_splitk_finish_kernel, dim3((B * Nout + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float *)workspace, KS, B, Nout, bias, relu, out);
    return pngpd_launch_status();
}

}  // extern "C"
