// libpngpd — training path of the per-point MLP trunk (batch-statistics BatchNorm + max-pool)
// as a sequence of passes over the cloud.  Two (B,N,128) tensors are kept between passes, both in a lane-major
// tile layout their consumers read without any shuffle: z2 = W2 h1 (written once by pass B, read by C, D, E) and the
// layer-2 gradient g2 (pass D -> pass E); h1 is recomputed from the 12-byte point, the 1024-channel layer is never
// stored.
//
// Replaces, in train mode, the autograd graph of
//   PointNetGPD/model/pointnet.py:29-33 (STN3d trunk) and :140-149 (PointNetfeat trunk)
// driven by PointNetGPD/main_1v.py:72-76 (forward, nll_loss, backward).
// The algebra (closed-form backward through conv1x1 -> BN(batch) -> [ReLU] -> max) is
// documented in DESIGN.md §"Training passes" and verified against autograd in fp64 by
// tests/train_algo_prototype.py.
//
// Per-channel affine forms used by every pass (so recomputed activations are bit-identical
// across passes):   h1 = relu(z1*s1c + t1c),  zhat1 = z1*is1 + nm1,   z1 = W1 x' + b1
//                   h2 = relu(z2*s2c + t2c),  zhat2 = z2*is2 + nm2,   z2 = W2 h1   (no bias)
#include "pngpd_tile.h"
#include "pngpd_glue_bodies.h"
#include "pngpd_internal.h"

#ifdef PNGPD_TIMING
__device__ unsigned long long pngpd_tm[16];
extern "C" int pngpd_tm_read(unsigned long long *host16, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host16, HIP_SYMBOL(pngpd_tm), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(pngpd_tm), z, sizeof(z)); }
    return 0;
}
#endif

struct TrainChan {
    const float *w1, *b1, *s1c, *t1c;   // layer 1: (64,3) raw, bias, scale, shift
    const float *w2p, *s2c, *t2c;       // layer 2: raw MFMA_B packed (128,64), scale, shift
    const u16 *w2x;                     // layer 2 as split_pack_bf16 fragments (NT > 0 kernels only)
};

// NT (template parameter of passes B, gather, D, E): 0 = exact fp32 on v_mfma_f32_32x32x2_f32 (the default, and the
// only arithmetic the parity tests pin); 1 / 3 = the contractions on bf16 / bf16x3 operands (pngpd_bf.h), everything
// else — BatchNorm statistics, masks, sums, accumulators — unchanged in fp32.

// Workgroups per cloud: the caller passes S explicitly (pngpd_trunk_splits() suggests one); nothing here is
// process-global, so buffer sizes computed by the caller and the launch always agree.
static inline bool splits_ok(int S, int T) { return S >= 1 && S <= T; }

// ---------------------------------------------------------------------------------------
// pass A: per-cloud input moments in fp64:  mom[b] = {sx,sy,sz, sxx,sxy,sxz, syy,syz,szz}
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cloud_moments_kernel(const float *__restrict__ x, int N,
                                                            double *__restrict__ mom) {
    cloud_moments_body(x, N, (int)blockIdx.x, mom);
}

// ---------------------------------------------------------------------------------------
// pass B: BN2 statistics.  part[blk][c][0..1] = sum, sum of squares of z2 = W2 h1 over the
// workgroup's valid points.
// ---------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256, 2) void trunk_bn2_stats_kernel(
    const float *__restrict__ x, int N, const float *__restrict__ trans, TrainChan P,
    int T, int S, float *__restrict__ part, f32x4 *__restrict__ z2t) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *h1 = smem;             // [TP][H1S]
    float *xs = h1 + TP * H1S;    // [3][TP]
    const Lane L;
    const int b = blockIdx.x / S, s = blockIdx.x - b * S;
    int t0, t1; tile_range(s, S, T, t0, t1);
    const float *xb = x + (size_t)b * 3 * N;
    float tm[9] = {0};
    const bool has_t = trans != nullptr;
    if (has_t) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tm[i] = trans[(size_t)b * 9 + i];
    }
    double sum = 0.0, sq = 0.0;
    f32x4 w2f[8];   // this wave's layer-2 weight fragments stay in registers for the whole kernel
    if (NT == 0) load_w2frag(w2f, P.w2p, L.wave, L);
    const L1C l1c = load_l1c(P.w1, P.b1, P.s1c, P.t1c, L);
    for (int tile = t0; tile < t1; ++tile) {
        stage_points(xb, N, tile, has_t, tm, xs, nullptr, L.tid);
        __syncthreads();
        layer1_tile(xs, l1c, h1, L);
        __syncthreads();
        f32x16 a0, a1;
        if constexpr (NT == 0) layer2_compute(h1, w2f, L, a0, a1);
        else layer2_compute_bf<NT>(h1, P.w2x, L.wave, L, a0, a1);
        const int nbase = tile * TP;
        if (z2t) {   // z2 is computed ONCE per step, here; passes C, D and E read it back (lane-major tiles, 512 B/point)
            if constexpr (NT == 1) {   // plain-bf16 mode: bf16 tiles, 256 B/point (pngpd_bf.h)
                bf_tile_store((uint4 *)z2t + ((size_t)(b * T + tile) * 4) * 256 + L.tid, a0, a1);
            } else {
            f32x4 *zt = z2t + ((size_t)(b * T + tile) * 8) * 256 + L.tid;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                zt[(size_t)rq * 256] = f32x4{a0[4 * rq], a0[4 * rq + 1], a0[4 * rq + 2], a0[4 * rq + 3]};
                zt[(size_t)(4 + rq) * 256] = f32x4{a1[4 * rq], a1[4 * rq + 1], a1[4 * rq + 2], a1[4 * rq + 3]};
            }
            }
        }
        // the tile's 32 values per lane as packed fp32 partials; the running sums over the workgroup's tiles are fp64
        // (a sequential fp32 accumulation over 512 values per lane was the largest round-off term of the batch
        // statistics; the fp64 adds are two instructions per tile)
        f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
        if (nbase + TP <= N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x2 v = {a0[r], a1[r]};
                s2 += v;
                q2 = __builtin_elementwise_fma(v, v, q2);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma_row(r, L.lane);
                const f32x2 v = {(nbase + row < N) ? a0[r] : 0.f, (nbase + 32 + row < N) ? a1[r] : 0.f};
                s2 += v;
                q2 = __builtin_elementwise_fma(v, v, q2);
            }
        }
        sum += (double)(s2[0] + s2[1]);
        sq += (double)(q2[0] + q2[1]);
        __syncthreads();   // h1/xs are rewritten by the next tile
    }
    sum += __shfl_xor(sum, 32);
    sq += __shfl_xor(sq, 32);
    if (L.h == 0) {
        float *o = part + ((size_t)blockIdx.x * 128 + L.wave * 32 + L.j) * 2;
        o[0] = (float)sum; o[1] = (float)sq;
    }
}

// ---------------------------------------------------------------------------------------
// pass C (main forward): z3s = (sgn*W3) h2 per point; per (cloud, channel) running max and
// argmax over the workgroup's tiles; per channel sum / sum of squares over valid points.
//   pmax/parg : [blk][1024]      psum : [blk][2][1024]
// ---------------------------------------------------------------------------------------
#define TRAIN_MAIN_LDS_FLOATS (TP * I1S + TP * I2S + 3 * TP + 4 * 1024)

// Same skeleton as trunk_infer_kernel (pngpd_trunk_infer.hip): unpadded XOR-swizzled LDS tiles, A fragments and
// layer-3 weight fragments double-buffered in registers, the tile's points fetched one tile ahead.
// LOADZ: z2 = W2 h1 was stored by pass B (z2t) and is read back — no points, no layer 1, no layer-2 MFMAs, no h1
// tile, two barriers per tile instead of three; the next tile's z2 is in flight during layer 3.
template <bool LOADZ>
__global__ __launch_bounds__(256, 2) void trunk_fwd_train_kernel(
    const float *__restrict__ x, int N, const float *__restrict__ trans, TrainChan P,
    const float *__restrict__ w3sp, int T, int S,
    float *__restrict__ pmax, int *__restrict__ parg, float *__restrict__ psum, float *__restrict__ psh,
    const f32x4 *__restrict__ z2t) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *h1 = smem;                 // [TP][I1S] swizzled
    float *h2 = h1 + TP * I1S;        // [TP][I2S] swizzled
    float *xs = h2 + TP * I2S;
    float *rm = xs + 3 * TP;          // [1024] running max
    int *ri = (int *)(rm + 1024);     // [1024] running argmax (point index)
    float *ss = (float *)(ri + 1024); // [1024] sum
    float *sq = ss + 1024;            // [1024] sum of squares
    const Lane L;
    wg_priority();
    const int b = blockIdx.x / S, s = blockIdx.x - b * S;
    int t0, t1; tile_range(s, S, T, t0, t1);
    const float *xb = x + (size_t)b * 3 * N;
    for (int i = L.tid; i < 1024; i += 256) { rm[i] = -INFINITY; ri[i] = 0; ss[i] = 0.f; sq[i] = 0.f; }
    f32x4 wa[16], wb[16];
    load_wfrag(wa, w3sp, L.wave, L);
    float tm[9] = {0};
    const bool has_t = trans != nullptr;
    if (has_t) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tm[i] = trans[(size_t)b * 9 + i];
    }
    float px0 = 0.f, px1 = 0.f, px2 = 0.f;
    if (!LOADZ && L.tid < TP) {
        int n = t0 * TP + L.tid; n = n < N ? n : N - 1;
        px0 = xb[n]; px1 = xb[N + n]; px2 = xb[2 * N + n];
    }
    const int cb2 = L.wave, c2 = cb2 * 32 + L.j;
    const float sc2 = P.s2c[c2], sh2 = P.t2c[c2];
    double hsum = 0.0;   // sum over this workgroup's valid points of h2[.][c2] (rows of this half-wave); fp64 across tiles
    f32x4 zq[8];
    if (LOADZ) {
        const f32x4 *zt = z2t + ((size_t)(b * T + t0) * 8) * 256 + L.tid;
#pragma unroll
        for (int i = 0; i < 8; ++i) zq[i] = zt[(size_t)i * 256];
    }
    TM_DECL
    for (int tile = t0; tile < t1; ++tile) {
        TM(0)
        if (LOADZ) {
            if (tile > t0) __syncthreads();   // every wave is done reading the previous tile's h2
            TM(1)
        } else if (L.tid < TP) {
            float x0 = px0, x1 = px1, x2 = px2;
            if (has_t) {
                x0 = fmaf(px2, tm[6], fmaf(px1, tm[3], px0 * tm[0]));
                x1 = fmaf(px2, tm[7], fmaf(px1, tm[4], px0 * tm[1]));
                x2 = fmaf(px2, tm[8], fmaf(px1, tm[5], px0 * tm[2]));
            }
            xs[L.tid] = x0; xs[TP + L.tid] = x1; xs[2 * TP + L.tid] = x2;
            if (tile + 1 < t1) {
                int n = (tile + 1) * TP + L.tid; n = n < N ? n : N - 1;
                px0 = xb[n]; px1 = xb[N + n]; px2 = xb[2 * N + n];
            }
        }
        if (!LOADZ) __syncthreads();
        if (!LOADZ) {   // layer 1 (3 -> 64) + batch-stat BN affine + ReLU, VALU: thread = (point = lane, 16-channel group = wave)
            const int p = L.lane;
            const float x0 = xs[p], x1 = xs[TP + p], x2 = xs[2 * TP + p];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = L.wave * 16 + g * 4 + e;   // wave-uniform -> scalar loads
                    const float z = fmaf(P.w1[c * 3 + 2], x2, fmaf(P.w1[c * 3 + 1], x1, fmaf(P.w1[c * 3], x0, P.b1[c])));
                    v[e] = fmaxf(fmaf(z, P.s1c[c], P.t1c[c]), 0.f);
                }
                *(f32x4 *)(h1 + swz(p, L.wave * 16 + g * 4, I1S)) = v;
            }
        }
        if (!LOADZ) __syncthreads();
        const int nbase = tile * TP;
        const bool full = nbase + TP <= N;
        {   // layer 2 (64 -> 128): read back (LOADZ) or on the MFMA
            f32x16 a0, a1;
            if (LOADZ) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { a0[r] = zq[r >> 2][r & 3]; a1[r] = zq[4 + (r >> 2)][r & 3]; }
                if (tile + 1 < t1) {
                    const f32x4 *zt = z2t + ((size_t)(b * T + tile + 1) * 8) * 256 + L.tid;
#pragma unroll
                    for (int i = 0; i < 8; ++i) zq[i] = zt[(size_t)i * 256];
                }
            } else {
                f32x4 w2f[8];
                load_w2frag(w2f, P.w2p, cb2, L);
                swz_compute<I1S, 8>(h1, w2f, L, a0, a1);
            }
            // column sums of h2 (the mean of h2 enters cvec of pass D and the closed-form dW3), finished HERE: left to
            // itself the compiler sinks the masked accumulation below the tile's eight layer-3 blocks and keeps all 32
            // activations in registers until then (32 VGPRs of a kernel at the 256 ceiling, ~200 VALU per tile).
            f32x2 hs2 = {0.f, 0.f};
            f32x2 hv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma_row(r, L.lane);
                hv[r] = f32x2{fmaxf(fmaf(a0[r], sc2, sh2), 0.f), fmaxf(fmaf(a1[r], sc2, sh2), 0.f)};
                h2[swz(row, c2, I2S)] = hv[r][0];
                h2[swz(32 + row, c2, I2S)] = hv[r][1];
            }
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) hs2 += hv[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mfma_row(r, L.lane);
                    hs2[0] += (nbase + row < N) ? hv[r][0] : 0.f;
                    hs2[1] += (nbase + 32 + row < N) ? hv[r][1] : 0.f;
                }
            }
            hsum += (double)(hs2[0] + hs2[1]);
            asm volatile("" : "+v"(hsum));
        }
        TM(2)
        __syncthreads();
        TM(3)
        auto reduce_block = [&](int cb, const f32x16 &a0, const f32x16 &a1) {
            // the block's running state is requested first: its LDS latency (queued behind the workgroup's A-fragment
            // reads) passes under the VALU work below instead of in four dependent round trips after it
            const int c = cb * 32 + L.j;
            const float rmc = rm[c], ssc = ss[c], sqc = sq[c];
            // first maximum over this lane's 32 rows (ascending row order) and the two moments
            float m, su, qu; int am;
            if (full) {
                lane_max_moments(a0, a1, m, am, su, qu);
                am += 4 * L.h;
            } else {
                m = a0[0]; am = mfma_row(0, L.lane);
#pragma unroll
                for (int r = 1; r < 16; ++r) { if (a0[r] > m) { m = a0[r]; am = mfma_row(r, L.lane); } }
#pragma unroll
                for (int r = 0; r < 16; ++r) { if (a1[r] > m) { m = a1[r]; am = 32 + mfma_row(r, L.lane); } }
                su = 0.f; qu = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mfma_row(r, L.lane);
                    const float v0 = (nbase + row < N) ? a0[r] : 0.f;
                    const float v1 = (nbase + 32 + row < N) ? a1[r] : 0.f;
                    su += v0 + v1; qu = fmaf(v0, v0, fmaf(v1, v1, qu));
                }
            }
            // the two row halves meet through v_permlane32_swap (no LDS round trip); ties go to the earlier row
            {
                float mlo, mhi; int alo, ahi;
                half_pair(m, mlo, mhi); half_pair(am, alo, ahi);
                const bool hi_wins = mhi > mlo || (mhi == mlo && ahi < alo);
                m = hi_wins ? mhi : mlo; am = hi_wins ? ahi : alo;
                su = half_sum(su); qu = half_sum(qu);
            }
            if (L.h == 0) {
                if (m > rmc) { rm[c] = m; int n = nbase + am; ri[c] = n < N ? n : N - 1; }
                ss[c] = ssc + su; sq[c] = sqc + qu;
            }
        };
#pragma unroll 1
        for (int cp = 0; cp < 4; ++cp) {
            const int cbA = L.wave + 8 * cp, cbB = cbA + 4, cbN = L.wave + ((8 * cp + 8) & 31);
            f32x16 a0, a1;
            load_wfrag(wb, w3sp, cbB, L);
            swz_compute<I2S, 16>(h2, wa, L, a0, a1);
            TM(4)
            reduce_block(cbA, a0, a1);
            TM(5)
            load_wfrag(wa, w3sp, cbN, L);
            swz_compute<I2S, 16>(h2, wb, L, a0, a1);
            TM(4)
            reduce_block(cbB, a0, a1);
            TM(5)
        }
        // no end-of-tile barrier: the next tile's xs/h1 writes do not alias h2, and the barrier before its layer 2
        // orders the h2 rewrite after every wave's layer-3 reads (as in trunk_infer_kernel).
    }
    TM_END
    hsum += __shfl_xor(hsum, 32);
    if (L.h == 0) {
        psh[(size_t)blockIdx.x * 128 + c2] = (float)hsum;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const int c = (L.wave + 4 * ci) * 32 + L.j;
            pmax[(size_t)blockIdx.x * 1024 + c] = rm[c];
            parg[(size_t)blockIdx.x * 1024 + c] = ri[c];
            psum[((size_t)blockIdx.x * 2) * 1024 + c] = ss[c];
            psum[((size_t)blockIdx.x * 2 + 1) * 1024 + c] = sq[c];
        }
    }
}

// ---------------------------------------------------------------------------------------
// gather pass: Gp[rng][c][k] = sum_{b in range} coef[b][c] * h2[b][k] evaluated at point idx[b][c]
// workgroup = (64-channel chunk cc, cloud range rng).
// ---------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256, 4) void trunk_bwd_gather_kernel(   // 4 workgroups per CU: <= 128 VGPRs (129 costs 25 %)
    const float *__restrict__ x, int B, int N, const float *__restrict__ trans, TrainChan P,
    const int *__restrict__ idx, const float *__restrict__ coef, int clouds_per_range,
    float *__restrict__ Gp, const ACvecArgs AT, int n_main) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x >= n_main) {   // tail workgroups (fused backward): A / cvec for pass D, row blockIdx.x - n_main
        a_cvec_finalize_body<256>(AT, (int)blockIdx.x - n_main, (double *)smem);
        return;
    }
    float *h1 = smem;
    float *xcf = h1 + TP * H1S;   // 2 x ([3][TP] points, [TP] coef): double-buffered by cloud parity
    const Lane L;
    const int cc = blockIdx.x & 15, rng = blockIdx.x >> 4;
    const int b0 = rng * clouds_per_range;
    const int b1 = (b0 + clouds_per_range < B) ? b0 + clouds_per_range : B;
    const bool has_t = trans != nullptr;
    // G accumulates in the MFMA layout of layer 2's output: lane (column j of channel block cb = wave, rows
    // mfma_row(r, lane) of both row halves) owns G[row][cb*32 + j] — h2 never goes through LDS, and a cloud costs two
    // barriers (points staged / h1 written) instead of four.
    f32x16 g0, g1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }
    const int cb = L.wave;
    const float sc = P.s2c[cb * 32 + L.j], sh = P.t2c[cb * 32 + L.j];
    f32x4 w2f[8];
    if (NT == 0) load_w2frag(w2f, P.w2p, L.wave, L);
    const L1C l1c = load_l1c(P.w1, P.b1, P.s1c, P.t1c, L);
    // the arg-max points of cloud b+1 (a dependent idx -> x gather) are fetched while cloud b is processed
    float nx0 = 0.f, nx1 = 0.f, nx2 = 0.f, ncf = 0.f;
    auto fetch = [&](int b) {
        if (L.tid < TP && b < b1) {
            const int c = cc * 64 + L.tid;
            const int n = idx[(size_t)b * 1024 + c];
            const float *xb = x + (size_t)b * 3 * N;
            nx0 = xb[n]; nx1 = xb[N + n]; nx2 = xb[2 * N + n];
            ncf = coef[(size_t)b * 1024 + c];
        }
    };
    fetch(b0);
    for (int b = b0; b < b1; ++b) {
        float *xs = xcf + ((b - b0) & 1) * 4 * TP, *cf = xs + 3 * TP;
        if (L.tid < TP) {
            float x0 = nx0, x1 = nx1, x2 = nx2;
            const float cfv = ncf;
            fetch(b + 1);
            if (has_t) {
                const float *tm = trans + (size_t)b * 9;
                float y0 = fmaf(x2, tm[6], fmaf(x1, tm[3], x0 * tm[0]));
                float y1 = fmaf(x2, tm[7], fmaf(x1, tm[4], x0 * tm[1]));
                float y2 = fmaf(x2, tm[8], fmaf(x1, tm[5], x0 * tm[2]));
                x0 = y0; x1 = y1; x2 = y2;
            }
            xs[L.tid] = x0; xs[TP + L.tid] = x1; xs[2 * TP + L.tid] = x2;
            cf[L.tid] = cfv;
        }
        __syncthreads();   // also: every wave has finished cloud b-1's layer 2 (the last reader of h1)
        layer1_tile(xs, l1c, h1, L);
        __syncthreads();
        f32x16 a0, a1;
        if constexpr (NT == 0) layer2_compute(h1, w2f, L, a0, a1);
        else layer2_compute_bf<NT>(h1, P.w2x, cb, L, a0, a1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma_row(r, L.lane);
            g0[r] = fmaf(cf[row], fmaxf(fmaf(a0[r], sc, sh), 0.f), g0[r]);
            g1[r] = fmaf(cf[32 + row], fmaxf(fmaf(a1[r], sc, sh), 0.f), g1[r]);
        }
        // no end-of-cloud barrier: xs/cf are double-buffered (cloud b+1 writes the other half; cloud b+2 rewrites this
        // half only after barrier 1 of cloud b+1, which every wave reaches after this epilogue)
    }
    float *o = Gp + ((size_t)rng * 1024 + cc * 64) * 128 + cb * 32 + L.j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = mfma_row(r, L.lane);
        o[(size_t)row * 128] = g0[r];
        o[(size_t)(32 + row) * 128] = g1[r];
    }
}

// ---------------------------------------------------------------------------------------
// pool refinement (the reduced-precision modes): zex[b][c] = (sgn W3)[c] . h2[b][:, idx[b][c]] in EXACT fp32
// arithmetic at the arg-max point the bf16 / bf16x3 pass C chose.
//   BatchNorm's mean / variance are averages over B*N products, where the bf16 product error largely averages out;
//   the pooled maxima are single values and carried that error in full (2^-9 .. 2^-16 relative) into the FC stacks,
//   whose batch-statistics BatchNorms amplify it on near-identical clouds.  After this pass the bf16 matrix pass
//   contributes only the CHOICE of the point.
// The arithmetic is the fp32 pass C's, operation for operation — x' = x^T T, layer 1 (layer1_tile), layer 2 on the fp32
// MFMA with the same fragment order (layer2_compute), h2 = relu(fma(z2, s2c, t2c)), and the layer-3 contraction as the
// same chain of v_mfma_f32_32x32x2_f32 over kb = 0..15, t = 0..3 on the sign-folded MFMA_B weights — so that wherever
// the chosen point IS the fp32 arg-max, the refined value is BIT-identical to the fp32 pass C's maximum (given the same
// BatchNorm-1/2 affine forms).  VARIANT 0 evaluates the 32x32 diagonal blocks on the matrix pipe (each wave of a pair
// owns (point block, channel block) = (w, w) and keeps its diagonal); VARIANTs 1-3 evaluate the 128-long chain on the
// VALU in the order the matrix instruction contracts its two k's (probed on the device by tests/test_gpu_refine.py).
// workgroup = (64-channel chunk cc, cloud range rng) like the gather pass.
// ---------------------------------------------------------------------------------------
#define REFINE_LDS_FLOATS (TP * H1S + TP * H2S + 8 * TP)

template <int VARIANT>
__global__ __launch_bounds__(256, 2) void trunk_pool_refine_kernel(
    const float *__restrict__ x, int B, int N, const float *__restrict__ trans, TrainChan P,
    const float *__restrict__ w3sp, const float *__restrict__ w3, const float *__restrict__ g3,
    const int *__restrict__ idx, int clouds_per_range, float *__restrict__ zex) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *h1 = smem;
    float *h2 = h1 + TP * H1S;    // [TP][H2S]
    float *xcf = h2 + TP * H2S;   // 2 x [3][TP] points (+ TP spare): double-buffered by cloud parity
    const Lane L;
    const int cc = blockIdx.x & 15, rng = blockIdx.x >> 4;
    const int b0 = rng * clouds_per_range;
    const int b1 = (b0 + clouds_per_range < B) ? b0 + clouds_per_range : B;
    const bool has_t = trans != nullptr;
    const int cb = L.wave, c2 = cb * 32 + L.j;
    const float sc = P.s2c[c2], sh = P.t2c[c2];
    f32x4 w2f[8];
    load_w2frag(w2f, P.w2p, L.wave, L);
    const L1C l1c = load_l1c(P.w1, P.b1, P.s1c, P.t1c, L);
    f32x4 w3f[16];   // VARIANT 0, waves 0 / 1: the sign-folded layer-3 fragments of channel block 2 cc + wave, resident
    if (VARIANT == 0 && L.wave < 2) load_wfrag(w3f, w3sp, cc * 2 + L.wave, L);
    // VARIANT > 0: lane p of a cloud's duty wave owns channel c = 64 cc + p.  Its sign-folded weight row (128 values)
    // stays in REGISTERS for the whole kernel (every wave holds it: the duty rotates over the SIMDs with the cloud) —
    // fetched per cloud from L2 it was 32 uncoalesced 16-byte loads (64 lines each) on the critical path of every
    // cloud: 293 us per launch at B = N = 1024 against 96 us for the gather pass that does the same layers 1-2.
    constexpr bool WREG = VARIANT == 1 || VARIANT == 2;     // (variant 3 is a probe only: it streams its row from L2)
    f32x4 wr[WREG ? 32 : 1];
    const f32x4 *wrow = (const f32x4 *)(w3 + (size_t)(cc * 64 + L.lane) * 128);
    const bool neg = VARIANT > 0 && g3[cc * 64 + L.lane] < 0.f;
    if constexpr (WREG) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { wr[i] = wrow[i]; if (neg) wr[i] = -wr[i]; }
    }
    float nx0 = 0.f, nx1 = 0.f, nx2 = 0.f;
    auto fetch = [&](int b) {
        if (L.tid < TP && b < b1) {
            const int n = idx[(size_t)b * 1024 + cc * 64 + L.tid];
            const float *xb = x + (size_t)b * 3 * N;
            nx0 = xb[n]; nx1 = xb[N + n]; nx2 = xb[2 * N + n];
        }
    };
    fetch(b0);
    for (int b = b0; b < b1; ++b) {
        float *xs = xcf + ((b - b0) & 1) * 4 * TP;
        if (L.tid < TP) {
            float x0 = nx0, x1 = nx1, x2 = nx2;
            fetch(b + 1);
            if (has_t) {
                const float *tm = trans + (size_t)b * 9;
                float y0 = fmaf(x2, tm[6], fmaf(x1, tm[3], x0 * tm[0]));
                float y1 = fmaf(x2, tm[7], fmaf(x1, tm[4], x0 * tm[1]));
                float y2 = fmaf(x2, tm[8], fmaf(x1, tm[5], x0 * tm[2]));
                x0 = y0; x1 = y1; x2 = y2;
            }
            xs[L.tid] = x0; xs[TP + L.tid] = x1; xs[2 * TP + L.tid] = x2;
        }
        __syncthreads();   // points staged; every wave is done with cloud b-1's h1 (layer 2) and h2 (layer 3 ends before it)
        layer1_tile(xs, l1c, h1, L);
        __syncthreads();
        f32x16 a0, a1;
        layer2_compute(h1, w2f, L, a0, a1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma_row(r, L.lane);
            h2[row * H2S + c2] = fmaxf(fmaf(a0[r], sc, sh), 0.f);
            h2[(32 + row) * H2S + c2] = fmaxf(fmaf(a1[r], sc, sh), 0.f);
        }
        __syncthreads();   // h2 complete
        float *o = zex + (size_t)b * 1024 + cc * 64;
        if constexpr (VARIANT == 0) {
            if (L.wave < 2) {
                const float *ap = h2 + (L.wave * 32 + L.j) * H2S + L.h * 4;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                f32x4 av = *(const f32x4 *)ap;
#pragma unroll
                for (int kb = 0; kb < 16; ++kb) {
                    f32x4 nv = av;
                    if (kb + 1 < 16) nv = *(const f32x4 *)(ap + (kb + 1) * 8);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc = mfma32(av[t], w3f[kb][t], acc);
                    av = nv;
                }
                // diagonal element (i == j) of the block: register r = (j & 3) + 4 (j >> 3) of the lane half h = (j >> 2) & 1
                const int rs = (L.j & 3) + 4 * (L.j >> 3);
                float dv = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) dv = (rs == r) ? acc[r] : dv;
                if (L.h == ((L.j >> 2) & 1)) o[L.wave * 32 + L.j] = dv;
            }
        } else {
            if (L.wave == ((b - b0) & 3)) {   // the duty wave rotates over the SIMDs with the cloud
                const float *hr = h2 + L.lane * H2S;
                float acc = 0.f;
                double accd = 0.0;
#pragma unroll
                for (int kb = 0; kb < 16; ++kb) {
                    const f32x4 alo = *(const f32x4 *)(hr + kb * 8), ahi = *(const f32x4 *)(hr + kb * 8 + 4);
                    f32x4 wlo, whi;
                    if constexpr (WREG) { wlo = wr[2 * kb]; whi = wr[2 * kb + 1]; }
                    else { wlo = wrow[2 * kb]; whi = wrow[2 * kb + 1]; if (neg) { wlo = -wlo; whi = -whi; } }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (VARIANT == 1) { acc = fmaf(alo[t], wlo[t], acc); acc = fmaf(ahi[t], whi[t], acc); }
                        else if (VARIANT == 2) { acc = fmaf(ahi[t], whi[t], acc); acc = fmaf(alo[t], wlo[t], acc); }
                        else {   // both products exact, one rounding per instruction
                            accd = (double)acc + (double)alo[t] * (double)wlo[t] + (double)ahi[t] * (double)whi[t];
                            acc = (float)accd;
                        }
                    }
                }
                o[L.lane] = acc;
            }
        }
        // no end-of-cloud barrier: xs is double-buffered; h1 / h2 are rewritten only behind the next cloud's barriers
    }
}

// ---------------------------------------------------------------------------------------
// pool refinement over the DISTINCT arg-max points of a cloud (round 6; the VALU variants 1 / 2 when the caller has the
// sign-folded MFMA_B weights at hand).  The 1,024 pooled channels of a cloud pick few different points — 53..98 of 1,024
// on the headline's iid clouds, 33..606 (mean 150) on diverse ones (profiles/r06_probe_unique_args.json) — but the
// kernel above evaluates the fp32 layers 1-2 at all B x 1,024 (cloud, channel) pairs: 17.6 GFLOP on the fp32 matrix
// pipe per launch, 0.18-0.22 ms, 10-16 % of a bf16-mode training step.  Here one workgroup owns a CLOUD: the arg-max
// points are marked in an N-bit LDS bitmap, a popcount scan numbers the distinct ones (ascending point index), layers
// 1-2 run on 64-point chunks of THAT list (the same layer1_tile / layer2_compute, so every h2 row has the bits the
// kernel above and the fp32 pass C give it), and each channel contracts its sign-folded weight row with the h2 row
// of its point on the VALU in the matrix instruction's order (VARIANT 1: k then k + 4; 2: k + 4 then k) — the weight
// row read from the MFMA_B fragments, where 32 consecutive channels' k-quads are contiguous (coalesced for lane =
// channel; the row-major weight the kernel above reads is not).  Same zex, bit for bit.
// ---------------------------------------------------------------------------------------
#define REFINE_DEDUP_LDS_FLOATS(W32) (TP * H1S + TP * H2S + 4 * TP + 2 * (W32) + 1024)

template <int VARIANT>
__global__ __launch_bounds__(256, 2) void trunk_pool_refine_dedup_kernel(
    const float *__restrict__ x, int B, int N, const float *__restrict__ trans, TrainChan P,
    const float *__restrict__ w3sp, const int *__restrict__ idx, int W32, float *__restrict__ zex) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *h1 = smem;                                  // [TP][H1S]
    float *h2 = h1 + TP * H1S;                         // [TP][H2S]
    float *xs = h2 + TP * H2S;                         // [3][TP] (+ TP spare)
    unsigned *bitmap = (unsigned *)(xs + 4 * TP);      // [W32] bit n: point n is some channel's arg-max
    int *pre = (int *)(bitmap + W32);                  // [W32] distinct points before word w
    int *ulist = pre + W32;                            // [<= 1024] the distinct points, ascending
    __shared__ int shw[4];
    const Lane L;
    const bool has_t = trans != nullptr;
    const int c2 = L.wave * 32 + L.j;
    const float sc = P.s2c[c2], sh = P.t2c[c2];
    f32x4 w2f[8];
    load_w2frag(w2f, P.w2p, L.wave, L);
    const L1C l1c = load_l1c(P.w1, P.b1, P.s1c, P.t1c, L);
    const int Wt = (W32 + 255) >> 8;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();                               // the previous cloud's lists are no longer read
        for (int w = L.tid; w < W32; w += 256) bitmap[w] = 0u;
        __syncthreads();
        int nq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            nq[q] = idx[(size_t)b * 1024 + L.tid + 256 * q];
            atomicOr(&bitmap[nq[q] >> 5], 1u << (nq[q] & 31));
        }
        __syncthreads();
        int loc = 0;
        for (int q = 0; q < Wt; ++q) { const int w = L.tid * Wt + q; if (w < W32) loc += __popc(bitmap[w]); }
        int incl = loc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (L.lane >= d) incl += t; }
        if (L.lane == 63) shw[L.wave] = incl;
        __syncthreads();
        int run = incl - loc, U = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < L.wave) run += shw[w]; U += shw[w]; }
        for (int q = 0; q < Wt; ++q) {
            const int w = L.tid * Wt + q;
            if (w < W32) {
                pre[w] = run;
                unsigned m = bitmap[w];
                while (m) { const int bit = __ffs(m) - 1; ulist[run++] = w * 32 + bit; m &= m - 1u; }
            }
        }
        __syncthreads();
        int slot[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            slot[q] = pre[nq[q] >> 5] + __popc(bitmap[nq[q] >> 5] & ((1u << (nq[q] & 31)) - 1u));
        const float *xb = x + (size_t)b * 3 * N;
        const int nch = (U + TP - 1) / TP;
        for (int ch = 0; ch < nch; ++ch) {
            if (L.tid < TP) {
                int u = ch * TP + L.tid; u = u < U ? u : U - 1;
                const int n = ulist[u];
                float x0 = xb[n], x1 = xb[N + n], x2 = xb[2 * N + n];
                if (has_t) {
                    const float *tm = trans + (size_t)b * 9;
                    const float y0 = fmaf(x2, tm[6], fmaf(x1, tm[3], x0 * tm[0]));
                    const float y1 = fmaf(x2, tm[7], fmaf(x1, tm[4], x0 * tm[1]));
                    const float y2 = fmaf(x2, tm[8], fmaf(x1, tm[5], x0 * tm[2]));
                    x0 = y0; x1 = y1; x2 = y2;
                }
                xs[L.tid] = x0; xs[TP + L.tid] = x1; xs[2 * TP + L.tid] = x2;
            }
            __syncthreads();   // points staged; every thread is done with the previous chunk's h1 / h2
            layer1_tile(xs, l1c, h1, L);
            __syncthreads();
            f32x16 a0, a1;
            layer2_compute(h1, w2f, L, a0, a1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma_row(r, L.lane);
                h2[row * H2S + c2] = fmaxf(fmaf(a0[r], sc, sh), 0.f);
                h2[(32 + row) * H2S + c2] = fmaxf(fmaf(a1[r], sc, sh), 0.f);
            }
            __syncthreads();   // h2 complete
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                if ((slot[q] >> 6) != ch) continue;
                const int c = L.tid + 256 * q;
                const float *hr = h2 + (slot[q] & 63) * H2S;
                // channel c's row in the MFMA_B fragments: quad (kb, h) of channel block c >> 5 at lane h * 32 + (c & 31)
                const f32x4 *wr = (const f32x4 *)w3sp + (size_t)((c >> 5) * 16) * 64 + (c & 31);
                float acc = 0.f;
#pragma unroll
                for (int kb = 0; kb < 16; ++kb) {
                    const f32x4 alo = *(const f32x4 *)(hr + kb * 8), ahi = *(const f32x4 *)(hr + kb * 8 + 4);
                    const f32x4 wlo = wr[kb * 64], whi = wr[kb * 64 + 32];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (VARIANT == 1) { acc = fmaf(alo[t], wlo[t], acc); acc = fmaf(ahi[t], whi[t], acc); }
                        else { acc = fmaf(ahi[t], whi[t], acc); acc = fmaf(alo[t], wlo[t], acc); }
                    }
                }
                zex[(size_t)b * 1024 + c] = acc;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// backward pass D: g2 = dL/d(bn2 output) per point; also the 128x128 second moments of h2 (the old separate
// "h-moments" pass: h2 is in LDS here anyway and this pass has idle MFMA slots).
//   dh2[point][k] = cvec[k] - (h2 A)[point][k] + sum_{c: idx[b][c]==point} coef[b][c] W3[c][k]
//   g2 = dh2 * (h2 > 0)
// Everything is a contraction on the MFMA, the sparse arg-extremum term included: a tile's hits (c, p) are
// compacted (ballot order = ascending channel, deterministic) into two lists by point half, and the term is
//   D[p][k] += sum_e  onehot[p][e] * (-coef[c_e]) * W3[c_e][k]
// i.e. extra k-steps of the SAME accumulators that hold h2 A: lane (row j, k-slot h) supplies
// A = (p_e == j) ? -coef : 0 for hit e = 2s+h and B = W3[c_e][32 cb + j] (a coalesced 128-B row piece from L2).
// No LDS scatter, no read-modify-write, no zero-fill.
//   g2t  [(b*T + tile)][8][256] float4 : lane-major hand-off to pass E (same (row, channel) ownership per lane
//        in both kernels: value v = 4*rq + e of thread tid is point block v>>4, mfma register v&15) — 8
//        coalesced 16-B stores per lane here, 8 coalesced 16-B loads there, no transposition through LDS.
//   pa   [blk][128][2] = sum g2, sum g2*zhat2
//   ps2  [blk][4 waves][3][16][64] : raw accumulators of the Gram blocks (w,w), (w,w+1 mod 4) and, in slot 2, the
//        partial of block (w,w+2) over the tile's first 32 points (w < 2) / of block (w-2,w) over its last 32 points
//        (w >= 2); the remaining blocks follow by symmetry (s2_at() in pngpd_train_glue.hip adds the two partials).
// ---------------------------------------------------------------------------------------
struct BwdDParams {
    const float *is2, *nm2;     // zhat2 = z2*is2 + nm2
    const float *Ap;            // (128,128) symmetric, MFMA_B packed
    const u16 *Ax;              // the same matrix as split_pack_bf16 fragments (NT > 0)
    const float *cvec;          // (128)
    const float *w3;            // (1024,128) raw row-major
    const int *idx;             // (B,1024)
    const float *coef;          // (B,1024)
};
#define BWD_D_HITS 1048   // per list: up to 1024 hit records + 24 zero records (16-hit steps, two steps of read-ahead)
#define BWD_D_LDS_FLOATS (TP * H2S + TP * H1S + 3 * TP + 1024 + 1024 + 4 * BWD_D_HITS + 16)
#define BWD_D_APL_KB 5     // k-blocks of A resident in LDS (fp32 kernel reading z2 back): 20 KB, 78.2 KB per workgroup
#define BWD_D_APL_FLOATS (BWD_D_APL_KB * 4 * 64 * 4)

// LOADZ: the raw layer-2 output z2 was stored by pass B (z2t, lane-major tiles) and is read back here instead of
// recomputing layers 1-2: 64 of the 324 MFMAs per wave and tile, the layer-1 VALU work, the h1 tile and one of the
// three barriers disappear, for 512 B per point of (overlapped) HBM reads.  !LOADZ (a caller without a z2t): recompute
// in fp32.  NT: 0 = fp32 MFMA; 1 / 3 = the three contractions on bf16 / bf16x3 operands (NT = 1 also keeps the z2 / g2
// tiles in bf16, pngpd_bf.h).
template <bool LOADZ, int NT>
__global__ __launch_bounds__(256, 2) void trunk_bwd_d_kernel(
    const float *__restrict__ x, int N, const float *__restrict__ trans, TrainChan P, BwdDParams D,
    int T, int S, const f32x4 *__restrict__ z2t, f32x4 *__restrict__ g2t, float *__restrict__ pa,
    float *__restrict__ ps2) {
    static_assert(NT == 0 || LOADZ, "the bf16 variants read z2 back");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *h2 = smem;                     // [TP][H2S], rows past N zeroed
    float *h1 = h2 + TP * H2S;            // [TP][H1S]  (not allocated when z2 is read back)
    float *xs = h1 + (LOADZ ? 0 : TP * H1S);   // [3][TP]
    float *cfl = xs + 3 * TP;             // [1024] coef row of this cloud
    int *idxl = (int *)(cfl + 1024);      // [1024] arg-extremum point of every channel of this cloud
    // [2 lists][BWD_D_HITS] hit records {-coef[c] as bits, (c << 5) | (point & 31)}: everything the sparse steps need in
    // ONE LDS read per hit (the old u16 list cost a second, dependent read of the coefficient row per hit, and every
    // LDS round trip of this phase queues behind the other waves' A-fragment reads)
    uint2 *hits = (uint2 *)(idxl + 1024);
    int *hcnt = (int *)(hits + 2 * BWD_D_HITS);               // [2 lists][4 waves] (16 words reserved)
    float *apl = (float *)(hcnt + 16);    // LOADZ && NT == 0: fragments of A, k-blocks [0, BWD_D_APL_KB), [kb][cb][lane] f32x4
    const Lane L;
    wg_priority();
    const int b = blockIdx.x / S, s = blockIdx.x - b * S;
    int t0, t1; tile_range(s, S, T, t0, t1);
    const float *xb = x + (size_t)b * 3 * N;
    float tm[9] = {0};
    const bool has_t = trans != nullptr;
    if (has_t) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tm[i] = trans[(size_t)b * 9 + i];
    }
    for (int i = L.tid; i < 1024; i += 256) {
        cfl[i] = D.coef[(size_t)b * 1024 + i];
        idxl[i] = D.idx[(size_t)b * 1024 + i];
    }
    if constexpr (LOADZ && NT == 0) {
        for (int e = L.tid; e < BWD_D_APL_KB * 4 * 64; e += 256) {
            const int lane = e & 63, cbb = (e >> 6) & 3, kb = e >> 8;
            ((f32x4 *)apl)[e] = ((const f32x4 *)D.Ap)[(size_t)(cbb * 16 + kb) * 64 + lane];
        }
    }
    double a1s = 0.0, a2s = 0.0;   // running sums over the workgroup's tiles (tile partials are fp32)
    const int cb = L.wave;
    const int c2 = cb * 32 + L.j;
    const float sc2 = P.s2c[c2], sh2 = P.t2c[c2], is2 = D.is2[c2], nm2 = D.nm2[c2], cv = D.cvec[c2];
    const unsigned long long ltmask = (1ull << L.lane) - 1ull;
    f32x16 gm0, gm1, gm2;   // Gram blocks (cb,cb), (cb,cb+1 mod 4), (cb,cb+2) [waves 0,1 only]
#pragma unroll
    for (int r = 0; r < 16; ++r) { gm0[r] = 0.f; gm1[r] = 0.f; gm2[r] = 0.f; }
    __syncthreads();   // cfl / idxl visible
    TM_DECL
    for (int tile = t0; tile < t1; ++tile) {
        const int nbase = tile * TP;
        f32x4 zq[8];
        uint4 zb[4];
        if (NT == 1) {            // bf16 tiles (plain-bf16 mode)
            const uint4 *zt = (const uint4 *)z2t + ((size_t)(b * T + tile) * 4) * 256 + L.tid;
#pragma unroll
            for (int i = 0; i < 4; ++i) zb[i] = zt[(size_t)i * 256];
        } else if (LOADZ) {
            const f32x4 *zt = z2t + ((size_t)(b * T + tile) * 8) * 256 + L.tid;
#pragma unroll
            for (int i = 0; i < 8; ++i) zq[i] = zt[(size_t)i * 256];
        } else {
            stage_points(xb, N, tile, has_t, tm, xs, nullptr, L.tid);
        }
        // hit census of this wave's channel quarter [256 wave, 256 wave + 256): ballots stay in scalar registers
        {
            int clo = 0, chi = 0;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int n = idxl[L.wave * 256 + it * 64 + L.lane] - nbase;
                clo += __popcll(__ballot(n >= 0 && n < 32));
                chi += __popcll(__ballot(n >= 32 && n < TP));
            }
            if (L.lane == 0) { hcnt[L.wave] = clo; hcnt[4 + L.wave] = chi; }
        }
        TM(0)
        __syncthreads();
        TM(1)
        f32x4 w2f[8];   // requested a phase ahead of layer 2 (not kept across the long MFMA phase: register budget)
        if (!LOADZ) {
            load_w2frag(w2f, P.w2p, cb, L);
            layer1_tile(xs, load_l1c(P.w1, P.b1, P.s1c, P.t1c, L), h1, L);
        }
        int nlo = 0, nhi = 0;
        {   // ordered compaction at the prefix offsets of the four quarters
            int olo = 0, ohi = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int a = hcnt[w], c = hcnt[4 + w];
                if (w < L.wave) { olo += a; ohi += c; }
                nlo += a; nhi += c;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int c = L.wave * 256 + it * 64 + L.lane;
                const int n = idxl[c] - nbase;
                const bool lo = n >= 0 && n < 32, hi = n >= 32 && n < TP;
                const unsigned long long mlo = __ballot(lo), mhi = __ballot(hi);
                const unsigned ncf = __float_as_uint(-cfl[c]);
                if (lo) hits[olo + __popcll(mlo & ltmask)] = uint2{ncf, (unsigned)((c << 5) | n)};
                if (hi) hits[BWD_D_HITS + ohi + __popcll(mhi & ltmask)] = uint2{ncf, (unsigned)((c << 5) | (n - 32))};
                olo += __popcll(mlo); ohi += __popcll(mhi);
            }
            // zero records behind both lists: the sparse steps read whole steps and two steps ahead, unguarded
            if (L.tid < 24) hits[nlo + L.tid] = uint2{0u, 0u};
            else if (L.tid >= 32 && L.tid < 56) hits[BWD_D_HITS + nhi + L.tid - 32] = uint2{0u, 0u};
        }
        nlo = __builtin_amdgcn_readfirstlane(nlo);
        nhi = __builtin_amdgcn_readfirstlane(nhi);
        f32x16 z0, z1;   // raw z2 of (this lane's rows, channel c2): ReLU mask and zhat2 are derived in the epilogue
        if (NT == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[8];
                bf_tile_unpack(zb[i], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { if (i < 2) z0[8 * i + e] = v[e]; else z1[8 * (i - 2) + e] = v[e]; }
            }
        } else if (LOADZ) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { z0[r] = zq[r >> 2][r & 3]; z1[r] = zq[4 + (r >> 2)][r & 3]; }
        } else {
            __syncthreads();   // h1 complete
            layer2_compute(h1, w2f, L, z0, z1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma_row(r, L.lane);
            h2[row * H2S + c2] = (nbase + row < N) ? fmaxf(fmaf(z0[r], sc2, sh2), 0.f) : 0.f;
            h2[(32 + row) * H2S + c2] = (nbase + 32 + row < N) ? fmaxf(fmaf(z1[r], sc2, sh2), 0.f) : 0.f;
        }
        TM(2)
        __syncthreads();   // h2 and the hit lists are complete
        TM(3)
        f32x16 d0, d1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
        if constexpr (NT > 0) {
            // --- the three contractions on bf16 / bf16x3 operands: 16 k indices per instruction ---
            k128_bf2<NT>(h2, D.Ax, cb, L, d0, d1);     // d = h2 A
            {   // sparse term: 16 hits per k-step; lane (j, h) supplies hits e0 + 8h .. 8h+7
                const float *w3c = D.w3 + c2;
                auto sparse = [&](const uint2 *hl, int n, f32x16 &d) {
#pragma unroll 1
                    for (int e0 = 0; e0 < n; e0 += 16) {
                        float av[8], bv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const uint2 rec = hl[e0 + 8 * L.h + u];   // zero records behind the list: coef 0, row 0
                            bv[u] = w3c[(size_t)(rec.y >> 5) * 128];
                            av[u] = ((int)(rec.y & 31) == L.j) ? __uint_as_float(rec.x) : 0.f;
                        }
                        f32x4 ah, al, bh, bl;
                        bf_pack8<NT>(av, ah, al);
                        bf_pack8<NT>(bv, bh, bl);
                        d = bf_mma<NT>(ah, al, bh, bl, d);
                    }
                };
                sparse(hits, nlo, d0);
                sparse(hits + BWD_D_HITS, nhi, d1);
            }
            {   // Gram: both operands are columns of h2 (k = point 16 st + 8h + u); the third block is split over the
                // points between waves cb and cb+2 exactly as in the fp32 loop below
                const float *colp = h2 + (8 * L.h) * H2S + L.j;
                const int o0 = cb * 32, o1 = ((cb + 1) & 3) * 32, o2 = ((cb + 2) & 3) * 32;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const bool third = (cb < 2) ? (st < 2) : (st >= 2);
                    float av[8], b1[8], b2[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float *rp = colp + (16 * st + u) * H2S;
                        av[u] = rp[o0]; b1[u] = rp[o1];
                        b2[u] = third ? rp[o2] : 0.f;
                    }
                    f32x4 ah, al, bh, bl;
                    bf_pack8<NT>(av, ah, al);
                    gm0 = bf_mma<NT>(ah, al, ah, al, gm0);
                    bf_pack8<NT>(b1, bh, bl);
                    gm1 = bf_mma<NT>(ah, al, bh, bl, gm1);
                    if (third) {
                        bf_pack8<NT>(b2, bh, bl);
                        gm2 = (cb < 2) ? bf_mma<NT>(ah, al, bh, bl, gm2) : bf_mma<NT>(bh, bl, ah, al, gm2);
                    }
                }
            }
        } else {
        {   // d = h2 A (K = 128).  (Explicitly software-pipelined variants of this loop, of the sparse loop and of the
            // Gram loop were measured: no gain, this pass is bound by its phase structure, not by operand latency.)
            const f32x4 *wp = (const f32x4 *)D.Ap + (size_t)(cb * 16) * 64 + L.lane;
            const float *a0p = h2 + L.j * H2S + L.h * 4;
            const float *a1p = h2 + (32 + L.j) * H2S + L.h * 4;
#ifndef PNGPD_DBG_NO_D_LDS
            constexpr bool apl_on = LOADZ;
#else
            constexpr bool apl_on = false;
#endif
            if constexpr (apl_on) {
                // k-blocks [0, BWD_D_APL_KB) of A from LDS, the rest through a 4-deep ring requested at the top (the LDS
                // blocks' 2,560 MFMA cycles cover the first round trip)
                const f32x4 *al = (const f32x4 *)apl + cb * 64 + L.lane;
                constexpr int NL = BWD_D_APL_KB, RING = 4;
                f32x4 wq[RING];
#pragma unroll
                for (int i = 0; i < RING; ++i) wq[i] = wp[(NL + i) * 64];
#pragma unroll
                for (int kb = 0; kb < 16; ++kb) {
                    f32x4 wv;
                    if (kb < NL) wv = al[kb * 256];
                    else {
                        wv = wq[(kb - NL) % RING];
                        if (kb + RING < 16) wq[(kb - NL) % RING] = wp[(kb + RING) * 64];
                    }
                    const f32x4 a0 = *(const f32x4 *)(a0p + kb * 8);
                    const f32x4 a1 = *(const f32x4 *)(a1p + kb * 8);
#pragma unroll
                    for (int t = 0; t < 4; ++t) { d0 = mfma32(a0[t], wv[t], d0); d1 = mfma32(a1[t], wv[t], d1); }
                }
            } else {
#pragma unroll 4
                for (int kb = 0; kb < 16; ++kb) {
                    const f32x4 wv = wp[kb * 64];
                    const f32x4 a0 = *(const f32x4 *)(a0p + kb * 8);
                    const f32x4 a1 = *(const f32x4 *)(a1p + kb * 8);
#pragma unroll
                    for (int t = 0; t < 4; ++t) { d0 = mfma32(a0[t], wv[t], d0); d1 = mfma32(a1[t], wv[t], d1); }
                }
            }
        }
        TM(4)
        // d -= sparse term: 8 hits (4 k-steps) per step.  The step is a chain LDS record -> W3 row piece (L2) -> MFMA,
        // and a tile has ~16 of them: left serial, the phase was 18,500 of a wave's 51,800 cycles per tile for 2,000
        // cycles of matrix work (tools/phase_times.py).  Software-pipelined two deep: the records of step i+2 and the
        // row pieces of step i+1 are in flight while step i is on the pipe.
        {
            const char *w3c = (const char *)(D.w3 + c2);
            auto sparse = [&](const uint2 *hl, int n, f32x16 &d) {
                if (n <= 0) return;
                const uint2 *hp = hl + L.h;
                uint2 r0[4], r1[4];
                float b0[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) r0[u] = hp[2 * u];
#pragma unroll
                for (int u = 0; u < 4; ++u) r1[u] = hp[8 + 2 * u];
#pragma unroll
                for (int u = 0; u < 4; ++u) b0[u] = *(const float *)(w3c + (size_t)(r0[u].y >> 5) * 512);
#pragma unroll 1
                for (int e0 = 0; e0 < n; e0 += 8) {
                    uint2 r2[4];
                    float b1[4], av[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) r2[u] = hp[e0 + 16 + 2 * u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) b1[u] = *(const float *)(w3c + (size_t)(r1[u].y >> 5) * 512);
#pragma unroll
                    for (int u = 0; u < 4; ++u) av[u] = ((int)(r0[u].y & 31) == L.j) ? __uint_as_float(r0[u].x) : 0.f;
#pragma unroll
                    for (int u = 0; u < 4; ++u) d = mfma32(av[u], b0[u], d);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { r0[u] = r1[u]; r1[u] = r2[u]; b0[u] = b1[u]; }
                }
            };
            sparse(hits, nlo, d0);
            sparse(hits + BWD_D_HITS, nhi, d1);
        }
        TM(5)
        // Gram of the tile: D[i][j] += A[i][k = point] B[k = point][j], both operands read column-wise from h2.
        // 10 blocks over 4 waves: every wave takes (cb,cb) and (cb,cb+1); the two remaining blocks (0,2) and (1,3) are
        // split over the POINTS — wave cb < 2 contracts the tile's first 32 points of block (cb,cb+2), wave cb+2 the
        // last 32 of the same block (s2_at() adds the two partial slots) — so each wave issues 80 MFMAs per tile
        // instead of 96 / 64.
        {
            const float *colp = h2 + L.h * H2S + L.j;
            const int o0 = cb * 32, o1 = ((cb + 1) & 3) * 32, o2 = ((cb + 2) & 3) * 32;
            auto gram = [&](int st0, int third) {   // third: 0 none, 1 block (cb, cb+2), 2 block (cb-2, cb)
#pragma unroll 4
                for (int st = st0; st < st0 + 16; ++st) {
                    const float *rp = colp + 2 * st * H2S;
                    const float av = rp[o0];
                    gm0 = mfma32(av, av, gm0); gm1 = mfma32(av, rp[o1], gm1);
                    if (third == 1) gm2 = mfma32(av, rp[o2], gm2);
                    if (third == 2) gm2 = mfma32(rp[o2], av, gm2);
                }
            };
            if (cb < 2) { gram(0, 1); gram(16, 0); } else { gram(0, 0); gram(16, 2); }
        }
        TM(6)
        }   // NT == 0
        // epilogue: g2 = (cvec - d) masked by ReLU(bn2) and validity; running sums; lane-major hand-off.
        // (Measured: writing this VALU work interleaved with the Gram MFMAs above — two k-steps, one element pair —
        // is SLOWER, 0.96 vs 0.89 ms: the partner wave on the SIMD already fills the MFMA pipe during the epilogue,
        // and fillers between the dependent Gram MFMAs only delay them.)
        {
            f32x4 *gt = g2t + ((size_t)(b * T + tile) * 8) * 256 + L.tid;
            f32x16 gb0, gb1;   // NT == 1: the tile leaves as bf16 (the sums below take the unrounded values)
            // per element: g = cv - d; mask = relu'(bn2(z)) (and validity in a ragged tile); a1 += g; a2 += g zhat2 —
            // on register pairs (packed subtract / FMAs), tile partials in fp32, running sums over the tiles in fp64
            const bool fullt = nbase + TP <= N;
            const f32x2 cv2 = {cv, cv}, sc22 = {sc2, sc2}, sh22 = {sh2, sh2}, is22 = {is2, is2}, nm22 = {nm2, nm2};
            f32x2 t1 = {0.f, 0.f}, t2 = {0.f, 0.f};
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 o0, o1;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const int r = rq * 4 + e;
                    const int row = mfma_row(r, L.lane);
                    const f32x2 zz0 = {z0[r], z0[r + 1]}, zz1 = {z1[r], z1[r + 1]};
                    f32x2 g0 = cv2 - f32x2{d0[r], d0[r + 1]}, g1 = cv2 - f32x2{d1[r], d1[r + 1]};
                    const f32x2 ac0 = __builtin_elementwise_fma(zz0, sc22, sh22), ac1 = __builtin_elementwise_fma(zz1, sc22, sh22);
                    if (fullt) {
                        g0[0] = ac0[0] > 0.f ? g0[0] : 0.f; g0[1] = ac0[1] > 0.f ? g0[1] : 0.f;
                        g1[0] = ac1[0] > 0.f ? g1[0] : 0.f; g1[1] = ac1[1] > 0.f ? g1[1] : 0.f;
                    } else {
                        g0[0] = (nbase + row < N && ac0[0] > 0.f) ? g0[0] : 0.f;
                        g0[1] = (nbase + row + 1 < N && ac0[1] > 0.f) ? g0[1] : 0.f;
                        g1[0] = (nbase + 32 + row < N && ac1[0] > 0.f) ? g1[0] : 0.f;
                        g1[1] = (nbase + 33 + row < N && ac1[1] > 0.f) ? g1[1] : 0.f;
                    }
                    t1 += g0 + g1;
                    t2 = __builtin_elementwise_fma(g0, __builtin_elementwise_fma(zz0, is22, nm22),
                         __builtin_elementwise_fma(g1, __builtin_elementwise_fma(zz1, is22, nm22), t2));
                    o0[e] = g0[0]; o0[e + 1] = g0[1]; o1[e] = g1[0]; o1[e + 1] = g1[1];
                    if (NT == 1) { gb0[r] = g0[0]; gb0[r + 1] = g0[1]; gb1[r] = g1[0]; gb1[r + 1] = g1[1]; }
                }
                if (NT != 1) {
                    gt[(size_t)rq * 256] = o0;
                    gt[(size_t)(4 + rq) * 256] = o1;
                }
            }
            a1s += (double)(t1[0] + t1[1]);
            a2s += (double)(t2[0] + t2[1]);
            if constexpr (NT == 1) bf_tile_store((uint4 *)g2t + ((size_t)(b * T + tile) * 4) * 256 + L.tid, gb0, gb1);
        }
        TM(7)
        // no end-of-tile barrier: the next tile's stage_points/census touch only xs/hcnt, whose readers all sit
        // before this tile's second barrier; h1, hits and h2 are rewritten after the next tile's first barrier.
    }
    TM_END
    a1s += __shfl_xor(a1s, 32);
    a2s += __shfl_xor(a2s, 32);
    if (L.h == 0) {
        float *o = pa + ((size_t)blockIdx.x * 128 + c2) * 2;
        o[0] = (float)a1s; o[1] = (float)a2s;
    }
    {
        float *o = ps2 + ((size_t)blockIdx.x * 12 + cb * 3) * 1024 + L.lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o[r * 64] = gm0[r];
            o[1024 + r * 64] = gm1[r];
            o[2048 + r * 64] = gm2[r];   // waves 2,3: the second-half partial of block (cb-2, cb)
        }
    }
}

// ---------------------------------------------------------------------------------------
// backward pass E: dz2 -> dh1 = W2^T dz2 -> g1 = dL/d(bn1 output); accumulates
//   pc [blk][64][2] = sum g1, sum g1*zhat1 ;  pR [blk][64][3] = sum_points g1 x^T (original x)
//   pW2 [blk][128][64] = sum_points dz2 h1^T  (= this workgroup's share of dL/dW2, contracted on the MFMA)
// g2t: pass D's lane-major hand-off (see there).
// ---------------------------------------------------------------------------------------
struct BwdEParams {
    const float *is1, *nm1;       // zhat1 = z1*is1 + nm1
    const float *is2, *nm2;
    const float *a1m, *a2m;       // (128) a1/M, a2/M
    const float *dsc2;            // (128) gamma2/sigma2
    const float *w2tp;            // W2^T as a (64,128) matrix, MFMA_B packed
    const u16 *w2tx;              // the same matrix as split_pack_bf16 fragments (NT > 0)
};
#define BWD_E_LDS_FLOATS (TP * H1S + TP * H2S + 12 * TP)

template <bool LOADZ, int NT>   // LOADZ: z2 read back from z2t instead of recomputing layer 2 (64 of 192 MFMAs per wave and tile)
__global__ __launch_bounds__(256, 2) void trunk_bwd_e_kernel(
    const float *__restrict__ x, int N, const float *__restrict__ trans, TrainChan P, BwdEParams E,
    int T, int S, const f32x4 *__restrict__ z2t, const f32x4 *__restrict__ g2t, float *__restrict__ pc,
    float *__restrict__ pR, float *__restrict__ pW2, const DW3Args WT, int n_main) {
    static_assert(NT == 0 || LOADZ, "the bf16 variants read z2 back");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x >= n_main) {   // tail workgroups (fused backward): dW3's finalize, block blockIdx.x - n_main
        dw3_finalize_body<2>(WT, (int)blockIdx.x - n_main, (double *)smem);
        return;
    }
    float *h1 = smem;
    float *dz = h1 + TP * H1S;    // [TP][H2S]
    float *xbuf = dz + TP * H2S;  // 2 x ([3][TP] transformed, [3][TP] original): double-buffered per tile parity
    float *w2l = xbuf + 12 * TP;  // W2^T fragments resident for the whole kernel (k128_lds / k128_bf_lds), K128_LDS_FLOATS
    const Lane L;
    wg_priority();
    if constexpr (NT == 0) k128_fill_lds(w2l, E.w2tp, L.tid);   // read after the first tile's barriers
    else k128_bf_fill_lds<NT>(w2l, E.w2tx, L.tid);
    const int b = blockIdx.x / S, s = blockIdx.x - b * S;
    int t0, t1; tile_range(s, S, T, t0, t1);
    const float *xb = x + (size_t)b * 3 * N;
    float tm[9] = {0};
    const bool has_t = trans != nullptr;
    if (has_t) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tm[i] = trans[(size_t)b * 9 + i];
    }
    const int cb = L.wave, c2 = cb * 32 + L.j;
    const float is2 = E.is2[c2], nm2 = E.nm2[c2], a1m = E.a1m[c2], a2m = E.a2m[c2], dsc = E.dsc2[c2];
    // dh1 tile owned by this wave: point block pb1, channel block cb1
    const int pb1 = L.wave >> 1, cb1 = L.wave & 1, c1 = cb1 * 32 + L.j;
    const float w10 = P.w1[c1 * 3], w11 = P.w1[c1 * 3 + 1], w12 = P.w1[c1 * 3 + 2], bb1 = P.b1[c1];
    const float is1 = E.is1[c1], nm1 = E.nm1[c1];
    double c1d = 0.0, c2d = 0.0, r0d = 0.0, r1d = 0.0, r2d = 0.0;   // running sums over the workgroup's tiles
    f32x16 pw0, pw1;   // dW2 rows o = cb*32 + i, columns {0,1}*32 + j
#pragma unroll
    for (int r = 0; r < 16; ++r) { pw0[r] = 0.f; pw1[r] = 0.f; }
    f32x4 w2f[8];   // this wave's layer-2 weight fragments stay in registers for the whole kernel
    if (!LOADZ) load_w2frag(w2f, P.w2p, cb, L);
    const L1C l1c = load_l1c(P.w1, P.b1, P.s1c, P.t1c, L);
    f32x4 gq[8], zq[8];
    uint4 gb[4], zb[4];   // NT == 1: the same values as bf16 tiles
    // The hand-off values of a tile are dead once its dz tile is written, so the next tile's can be requested into the
    // same registers a tile ahead.  WHERE matters: vmcnt retires in order, so any later load that is waited on (the
    // W2^T fragment ring of the first contraction) would wait for these HBM reads as well — requested right after the dz
    // tile they cost 0.412 -> 0.528 ms.  They are issued after the last fragment of the first contraction instead; the
    // g1 epilogue, the dW2 contraction (LDS operands only) and the next tile's staging cover the round trip.
    auto fetch_tile = [&](int tile) {
        if (NT == 1) {
            const uint4 *gt = (const uint4 *)g2t + ((size_t)(b * T + tile) * 4) * 256 + L.tid;
            const uint4 *zt = (const uint4 *)z2t + ((size_t)(b * T + tile) * 4) * 256 + L.tid;
#pragma unroll
            for (int i = 0; i < 4; ++i) { gb[i] = gt[(size_t)i * 256]; zb[i] = zt[(size_t)i * 256]; }
        } else {
            const f32x4 *gt = g2t + ((size_t)(b * T + tile) * 8) * 256 + L.tid;
#pragma unroll
            for (int i = 0; i < 8; ++i) gq[i] = gt[(size_t)i * 256];
            if (LOADZ) {
                const f32x4 *zt = z2t + ((size_t)(b * T + tile) * 8) * 256 + L.tid;
#pragma unroll
                for (int i = 0; i < 8; ++i) zq[i] = zt[(size_t)i * 256];
            }
        }
    };
    fetch_tile(t0);
    // the tile's points travel the same way: requested at the top of the previous tile, written to the other parity of
    // xbuf just before that tile's hand-off prefetch (no vmem wait is left behind the HBM reads)
    float px0 = 0.f, px1 = 0.f, px2 = 0.f;
    auto load_points = [&](int tile) {
        if (L.tid < TP) {
            int n = tile * TP + L.tid; n = n < N ? n : N - 1;
            px0 = xb[n]; px1 = xb[N + n]; px2 = xb[2 * N + n];
        }
    };
    auto store_points = [&](int tile) {
        if (L.tid < TP) {
            float *xs_ = xbuf + ((tile - t0) & 1) * 6 * TP, *xo_ = xs_ + 3 * TP;
            xo_[L.tid] = px0; xo_[TP + L.tid] = px1; xo_[2 * TP + L.tid] = px2;
            float y0 = px0, y1 = px1, y2 = px2;
            if (has_t) {
                y0 = fmaf(px2, tm[6], fmaf(px1, tm[3], px0 * tm[0]));
                y1 = fmaf(px2, tm[7], fmaf(px1, tm[4], px0 * tm[1]));
                y2 = fmaf(px2, tm[8], fmaf(px1, tm[5], px0 * tm[2]));
            }
            xs_[L.tid] = y0; xs_[TP + L.tid] = y1; xs_[2 * TP + L.tid] = y2;
        }
    };
    load_points(t0);
    store_points(t0);
    TM_DECL
    for (int tile = t0; tile < t1; ++tile) {
        const int nbase = tile * TP;
        float *xs = xbuf + ((tile - t0) & 1) * 6 * TP, *xo = xs + 3 * TP;
        if (tile + 1 < t1) load_points(tile + 1);
        TM(0)
        __syncthreads();
        TM(1)
        layer1_tile(xs, l1c, h1, L);
        TM(2)
        if (!LOADZ) __syncthreads();   // layer 2 reads h1; with LOADZ the dz tile below depends on registers only
        {
            f32x16 a0, a1, gv0, gv1;
            if (NT == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float vz[8], vg[8];
                    bf_tile_unpack(zb[i], vz);
                    bf_tile_unpack(gb[i], vg);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (i < 2) { a0[8 * i + e] = vz[e]; gv0[8 * i + e] = vg[e]; }
                        else { a1[8 * (i - 2) + e] = vz[e]; gv1[8 * (i - 2) + e] = vg[e]; }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { gv0[r] = gq[r >> 2][r & 3]; gv1[r] = gq[4 + (r >> 2)][r & 3]; }
                if (LOADZ) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { a0[r] = zq[r >> 2][r & 3]; a1[r] = zq[4 + (r >> 2)][r & 3]; }
                } else {
                    layer2_compute(h1, w2f, L, a0, a1);
                }
            }
            // dz2 = dsc (g2 - a1m - zhat2 a2m), zhat2 = z2 is2 + nm2 — per element exactly the operations pass D used
            // for its sums of g2 zhat2 (folding the constants into one affine map of (g2, z2) is cheaper but gives a
            // zhat2 that differs from pass D's in the last bit of its CONSTANTS: a systematic error that adds up
            // coherently over the 10^6 points of dW2 = sum dz2 h1^T — 20x its round-off, measured).  On point pairs.
            if (nbase + TP <= N) {
                const f32x2 is22 = {is2, is2}, nm22 = {nm2, nm2}, a1m2 = {a1m, a1m}, na2m2 = {-a2m, -a2m}, dsc2 = {dsc, dsc};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int row = mfma_row(r, L.lane);
                    const f32x2 zh0 = __builtin_elementwise_fma(f32x2{a0[r], a0[r + 1]}, is22, nm22);
                    const f32x2 zh1 = __builtin_elementwise_fma(f32x2{a1[r], a1[r + 1]}, is22, nm22);
                    const f32x2 d0 = dsc2 * __builtin_elementwise_fma(zh0, na2m2, f32x2{gv0[r], gv0[r + 1]} - a1m2);
                    const f32x2 d1 = dsc2 * __builtin_elementwise_fma(zh1, na2m2, f32x2{gv1[r], gv1[r + 1]} - a1m2);
                    dz[row * H2S + c2] = d0[0]; dz[(row + 1) * H2S + c2] = d0[1];
                    dz[(32 + row) * H2S + c2] = d1[0]; dz[(33 + row) * H2S + c2] = d1[1];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mfma_row(r, L.lane);
                    const bool v0 = nbase + row < N, v1 = nbase + 32 + row < N;
                    const float z0 = fmaf(a0[r], is2, nm2), z1 = fmaf(a1[r], is2, nm2);
                    dz[row * H2S + c2] = v0 ? dsc * fmaf(z0, -a2m, gv0[r] - a1m) : 0.f;
                    dz[(32 + row) * H2S + c2] = v1 ? dsc * fmaf(z1, -a2m, gv1[r] - a1m) : 0.f;
                }
            }
        }
        TM(3)
        __syncthreads();
        TM(4)
        {
            // dh1[point][c1] = sum_o dz[point][o] * W2[o][c1]   (K = 128), one 32x32 tile per wave
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#ifndef PNGPD_DBG_NO_E_LDS
            if constexpr (NT == 0) k128_lds(dz, w2l, E.w2tp, cb1, pb1, L, acc);
#else
            f32x16 unused;
            if constexpr (NT == 0) k128_stream<1>(dz, E.w2tp, cb1, pb1, L, acc, unused);
#endif
            else k128_bf_lds<NT>(dz, w2l, E.w2tx, cb1, pb1, L, acc);
            if (tile + 1 < t1) { store_points(tile + 1); fetch_tile(tile + 1); }
            TM(5)
            // g1 = dh1 masked by ReLU(bn1) (rows past N: dz == 0 -> 0); c1 = sum g1, c2 = sum g1 zhat1, R = sum g1 x^T,
            // all on point PAIRS (packed adds / FMAs; zhat1 through the same FMA chain as layer 1)
            const f32x2 w102 = {w10, w10}, w112 = {w11, w11}, w122 = {w12, w12}, bb12 = {bb1, bb1};
            const f32x2 is12 = {is1, is1}, nm12 = {nm1, nm1};
            f32x2 cs2 = {0.f, 0.f}, cc2 = {0.f, 0.f}, rr0 = {0.f, 0.f}, rr1 = {0.f, 0.f}, rr2 = {0.f, 0.f};   // even / odd points
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int pt = pb1 * 32 + mfma_row(4 * rq, L.lane);   // 4 consecutive points
                const f32x4 q0 = *(const f32x4 *)(xo + pt), q1 = *(const f32x4 *)(xo + TP + pt),
                            q2 = *(const f32x4 *)(xo + 2 * TP + pt);
                const f32x4 y0 = *(const f32x4 *)(xs + pt), y1 = *(const f32x4 *)(xs + TP + pt),
                            y2 = *(const f32x4 *)(xs + 2 * TP + pt);
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const int r = 4 * rq + e;
                    const f32x2 g = {(h1[(pt + e) * H1S + c1] > 0.f) ? acc[r] : 0.f,
                                     (h1[(pt + e + 1) * H1S + c1] > 0.f) ? acc[r + 1] : 0.f};
                    const f32x2 z1 = __builtin_elementwise_fma(w122, f32x2{y2[e], y2[e + 1]},
                                     __builtin_elementwise_fma(w112, f32x2{y1[e], y1[e + 1]},
                                     __builtin_elementwise_fma(w102, f32x2{y0[e], y0[e + 1]}, bb12)));
                    cs2 += g;
                    cc2 = __builtin_elementwise_fma(g, __builtin_elementwise_fma(z1, is12, nm12), cc2);
                    rr0 = __builtin_elementwise_fma(g, f32x2{q0[e], q0[e + 1]}, rr0);
                    rr1 = __builtin_elementwise_fma(g, f32x2{q1[e], q1[e + 1]}, rr1);
                    rr2 = __builtin_elementwise_fma(g, f32x2{q2[e], q2[e + 1]}, rr2);
                }
            }
            c1d += (double)(cs2[0] + cs2[1]); c2d += (double)(cc2[0] + cc2[1]);
            r0d += (double)(rr0[0] + rr0[1]); r1d += (double)(rr1[0] + rr1[1]); r2d += (double)(rr2[0] + rr2[1]);
        }
        TM(6)
        // dW2 += dz^T h1 : contraction over the tile's 64 points (rows past N carry dz == 0)
        if constexpr (NT > 0) {
            const float *dzc = dz + (8 * L.h) * H2S + cb * 32 + L.j, *h1c = h1 + (8 * L.h) * H1S + L.j;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                float av[8], b0[8], b1[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    av[u] = dzc[(16 * st + u) * H2S];
                    b0[u] = h1c[(16 * st + u) * H1S]; b1[u] = h1c[(16 * st + u) * H1S + 32];
                }
                f32x4 ah, al, bh, bl;
                bf_pack8<NT>(av, ah, al);
                bf_pack8<NT>(b0, bh, bl);
                pw0 = bf_mma<NT>(ah, al, bh, bl, pw0);
                bf_pack8<NT>(b1, bh, bl);
                pw1 = bf_mma<NT>(ah, al, bh, bl, pw1);
            }
        } else {
            const float *dzc = dz + L.h * H2S + cb * 32 + L.j, *h1c = h1 + L.h * H1S + L.j;
            float ca[4], c0[4], c1v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { ca[u] = dzc[2 * u * H2S]; c0[u] = h1c[2 * u * H1S]; c1v[u] = h1c[2 * u * H1S + 32]; }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float na[4], n0[4], n1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    na[u] = ca[u]; n0[u] = c0[u]; n1[u] = c1v[u];
                    if (g + 1 < 8) {
                        const int st = (g + 1) * 4 + u;
                        na[u] = dzc[2 * st * H2S]; n0[u] = h1c[2 * st * H1S]; n1[u] = h1c[2 * st * H1S + 32];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    pw0 = mfma32(ca[u], c0[u], pw0);
                    pw1 = mfma32(ca[u], c1v[u], pw1);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { ca[u] = na[u]; c0[u] = n0[u]; c1v[u] = n1[u]; }
            }
        }
        TM(7)
        // no end-of-tile barrier: xs/xo are double-buffered; h1 and dz are rewritten only after the next tile's
        // first barrier, which every wave reaches after finishing this tile.
    }
    TM_END
    {
        float *oW = pW2 + (size_t)blockIdx.x * 128 * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = cb * 32 + mfma_row(r, L.lane);
            oW[o * 64 + L.j] = pw0[r];
            oW[o * 64 + 32 + L.j] = pw1[r];
        }
    }
    c1d += __shfl_xor(c1d, 32); c2d += __shfl_xor(c2d, 32);
    r0d += __shfl_xor(r0d, 32); r1d += __shfl_xor(r1d, 32); r2d += __shfl_xor(r2d, 32);
    const float c1s = (float)c1d, c2s = (float)c2d, r0 = (float)r0d, r1 = (float)r1d, r2 = (float)r2d;
    // two waves (pb1 = 0,1) own the same channel block: combine through LDS
    __syncthreads();   // every wave is done with dz
    float *red = dz;
    if (L.h == 0) {
        float *o = red + (L.wave * 32 + L.j) * 5;
        o[0] = c1s; o[1] = c2s; o[2] = r0; o[3] = r1; o[4] = r2;
    }
    __syncthreads();
    if (L.tid < 64) {
        const int cb_ = L.tid >> 5, jj = L.tid & 31;
        const float *p0 = red + ((0 * 2 + cb_) * 32 + jj) * 5;   // wave = pb1*2 + cb1
        const float *p1 = red + ((1 * 2 + cb_) * 32 + jj) * 5;
        float *oc = pc + ((size_t)blockIdx.x * 64 + L.tid) * 2;
        oc[0] = p0[0] + p1[0]; oc[1] = p0[1] + p1[1];
        float *oR = pR + ((size_t)blockIdx.x * 64 + L.tid) * 3;
        oR[0] = p0[2] + p1[2]; oR[1] = p0[3] + p1[3]; oR[2] = p0[4] + p1[4];
    }
}

#include "pngpd_bwd_bf.h"   // round 6: passes D / E of the bf16 modes, own structure

// ---------------------------------------------------------------------------------------
// BatchNorm1d over the batch (FC stacks) — train forward / backward, optional fused ReLU.
// block = 16 channels x 64 row lanes (1024 threads); thread (cx, ry) owns rows ry, ry+64, ...  With REG (B <= 1024)
// the thread's <= 16 values are loaded ONCE, all loads in flight together, and stay in registers through the
// statistics and the normalisation (the layer is 2 MB: the kernel is pure load latency, so one exposed round trip
// instead of three, and 2x the workgroups of a 32-channel block).
// ---------------------------------------------------------------------------------------
#define BN1D_CW 16
#define BN1D_RL (PNGPD_ASAN ? 16 : 64)   // row lanes; the sanitizer build runs 256-thread workgroups (pngpd_common.h)
#define BN1D_NV 16

__device__ __forceinline__ float bn1d_colsum(float (*red)[BN1D_CW + 1], int cx, int ry, float v) {
    __syncthreads();          // previous use of red finished
    red[ry][cx] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll 8
    for (int i = 0; i < BN1D_RL; ++i) t += red[i][cx];
    return t;
}

template <bool REG>
__global__ __launch_bounds__(BN1D_CW * BN1D_RL) void bn1d_fwd_train_kernel(
    const float *__restrict__ z, int B, int C, const float *__restrict__ gamma,
    const float *__restrict__ beta, float eps, int relu, float *__restrict__ y,
    float *__restrict__ mean_out, float *__restrict__ var_out,
    float momentum, float *rm, float *rv, long long *nbt) {
    __shared__ float red[BN1D_RL][BN1D_CW + 1];
    const int cx = threadIdx.x & (BN1D_CW - 1), ry = threadIdx.x / BN1D_CW;
    const int c = blockIdx.x * BN1D_CW + cx;
    const bool ok = c < C;
    const float *zc = z + (ok ? c : 0);
    float zv[BN1D_NV];
    float s = 0.f;
    if (REG) {
#pragma unroll
        for (int i = 0; i < BN1D_NV; ++i) {
            const int b = ry + BN1D_RL * i;
            zv[i] = (ok && b < B) ? zc[(size_t)b * C] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < BN1D_NV; ++i) s += zv[i];
    } else if (ok) {
        for (int b = ry; b < B; b += BN1D_RL) s += zc[(size_t)b * C];
    }
    const float mean = bn1d_colsum(red, cx, ry, s) / (float)B;
    float q = 0.f;
    if (REG) {
#pragma unroll
        for (int i = 0; i < BN1D_NV; ++i) {
            const float d = (ry + BN1D_RL * i < B) ? zv[i] - mean : 0.f;
            q = fmaf(d, d, q);
        }
    } else if (ok) {
        for (int b = ry; b < B; b += BN1D_RL) { const float d = zc[(size_t)b * C] - mean; q = fmaf(d, d, q); }
    }
    const float var = bn1d_colsum(red, cx, ry, q) / (float)B;
    if (!ok) return;
    const float inv = 1.0f / sqrtf(var + eps);
    const float g = gamma[c], be = beta[c];
    float *yc = y + c;
    if (REG) {
#pragma unroll
        for (int i = 0; i < BN1D_NV; ++i) {
            const int b = ry + BN1D_RL * i;
            float v = (zv[i] - mean) * inv * g + be;
            if (relu) v = v < 0.f ? 0.f : v;
            if (b < B) yc[(size_t)b * C] = v;
        }
    } else {
        for (int b = ry; b < B; b += BN1D_RL) {
            float v = (zc[(size_t)b * C] - mean) * inv * g + be;
            if (relu) v = v < 0.f ? 0.f : v;
            yc[(size_t)b * C] = v;
        }
    }
    if (ry == 0) {
        mean_out[c] = mean; var_out[c] = var;
        if (rm) {   // nn.BatchNorm1d running statistics: momentum, unbiased variance
            rm[c] = (1.f - momentum) * rm[c] + momentum * mean;
            rv[c] = (1.f - momentum) * rv[c] + momentum * var * ((float)B / (float)(B > 1 ? B - 1 : 1));
        }
        if (nbt && c == 0) *nbt += 1;
    }
}

// dy: gradient wrt the (post-ReLU if relu) output y.  dz, dgamma, dbeta out.
template <bool REG>
__global__ __launch_bounds__(BN1D_CW * BN1D_RL) void bn1d_bwd_kernel(
    const float *__restrict__ dy, const float *__restrict__ z, const float *__restrict__ y, int B, int C,
    const float *__restrict__ gamma, const float *__restrict__ mean, const float *__restrict__ var,
    float eps, int relu, float *__restrict__ dz, float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ float red[BN1D_RL][BN1D_CW + 1];
    const int cx = threadIdx.x & (BN1D_CW - 1), ry = threadIdx.x / BN1D_CW;
    const int c = blockIdx.x * BN1D_CW + cx;
    const bool ok = c < C;
    const int cc = ok ? c : 0;
    const float mu = mean[cc], inv = 1.0f / sqrtf(var[cc] + eps);
    float gv[BN1D_NV], xh[BN1D_NV];
    float s1 = 0.f, s2 = 0.f;
    if (REG) {
        float yv[BN1D_NV];
#pragma unroll
        for (int i = 0; i < BN1D_NV; ++i) {
            const int b = ry + BN1D_RL * i;
            const bool in = ok && b < B;
            const unsigned o = (unsigned)b * (unsigned)C + (unsigned)cc;   // B <= 1024 here: B*C fits 32 bits
            gv[i] = in ? dy[o] : 0.f;
            xh[i] = in ? z[o] : mu;
            yv[i] = (in && relu) ? y[o] : 1.f;
        }
#pragma unroll
        for (int i = 0; i < BN1D_NV; ++i) {
            if (!(yv[i] > 0.f)) gv[i] = 0.f;
            xh[i] = (xh[i] - mu) * inv;
            s1 += gv[i];
            s2 = fmaf(gv[i], xh[i], s2);
        }
    } else if (ok) {
        for (int b = ry; b < B; b += BN1D_RL) {
            const size_t i = (size_t)b * C + c;
            float g = dy[i];
            if (relu && !(y[i] > 0.f)) g = 0.f;
            s1 += g;
            s2 = fmaf(g, (z[i] - mu) * inv, s2);
        }
    }
    const float t1 = bn1d_colsum(red, cx, ry, s1);
    const float t2 = bn1d_colsum(red, cx, ry, s2);
    if (!ok) return;
    const float gi = gamma[c] * inv, m1 = t1 / (float)B, m2 = t2 / (float)B;
    if (REG) {
#pragma unroll
        for (int i = 0; i < BN1D_NV; ++i) {
            const int b = ry + BN1D_RL * i;
            if (b < B) dz[(size_t)b * C + c] = gi * (gv[i] - m1 - xh[i] * m2);
        }
    } else {
        for (int b = ry; b < B; b += BN1D_RL) {
            const size_t i = (size_t)b * C + c;
            float g = dy[i];
            if (relu && !(y[i] > 0.f)) g = 0.f;
            dz[i] = gi * (g - m1 - (z[i] - mu) * inv * m2);
        }
    }
    if (ry == 0) { dgamma[c] = t2; dbeta[c] = t1; }
}

// dlogits = g - exp(logp) * rowsum(g)      (backward of F.log_softmax, pointnet.py:194)
__global__ void log_softmax_bwd_kernel(const float *__restrict__ g, const float *__restrict__ logp,
                                       int B, int K, float *__restrict__ dlogits) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += g[(size_t)b * K + k];
    for (int k = 0; k < K; ++k) dlogits[(size_t)b * K + k] = g[(size_t)b * K + k] - expf(logp[(size_t)b * K + k]) * s;
}

// F.nll_loss(logp, target) (main_1v.py:74): one workgroup, fp64 accumulation in a fixed order (deterministic).
__global__ __launch_bounds__(256) void nll_fwd_kernel(const float *__restrict__ logp, const long long *__restrict__ target,
                                                      int B, int K, int mean, float *__restrict__ loss) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        const long long t = target[b];
        if (t >= 0 && t < K) acc -= (double)logp[(size_t)b * K + t];
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) acc += __shfl_xor(acc, k);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = ((sh[0] + sh[1]) + sh[2]) + sh[3];
        *loss = (float)(mean ? tot / (double)B : tot);
    }
}

// Backward of log_softmax with F.nll_loss's backward folded into its upstream: the loss path contributes
// -(gloss / B) (ATen's own expression, a division) at the target column; an explicit upstream g on the log-probabilities
// is added when present.  Same per-element expression as log_softmax_bwd_kernel, so the two-launch form (ATen's
// nll_loss_backward, then log_softmax_bwd) and this one agree bit for bit.
__global__ void nll_log_softmax_bwd_kernel(const float *__restrict__ g, const float *__restrict__ gloss,
                                           const long long *__restrict__ target, const float *__restrict__ logp,
                                           int B, int K, int mean, float *__restrict__ dlogits) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const long long t = target[b];
    const float gl = mean ? -(*gloss / (float)B) : -*gloss;
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
        float gk = g ? g[(size_t)b * K + k] : 0.f;
        if (k == t) gk = g ? gk + gl : gl;
        s += gk;
    }
    for (int k = 0; k < K; ++k) {
        float gk = g ? g[(size_t)b * K + k] : 0.f;
        if (k == t) gk = g ? gk + gl : gl;
        dlogits[(size_t)b * K + k] = gk - expf(logp[(size_t)b * K + k]) * s;
    }
}


// ---------------------------------------------------------------------------------------
// Backward of a Linear layer y = x W^T + b (pointnet.py:35-37,191-193) in ONE launch, operands read in place
// (no transposed copies): for upstream g (B,Nout)
//   dW (Nout,K) = g^T x   contraction over the batch   — A[i = n][kk = b] = g[b][n], B[kk = b][j = k] = x[b][k]:
//                         both operands are read along rows, i.e. coalesced exactly as the MFMA wants them
//   dx (B,K)    = g W     contraction over Nout        — A[i = b][kk = n] = g[b][n] (float4 along n when
//                         Nout % 8 == 0), B[kk = n][j = k] = W[n][k]
//   db (Nout)   = sum_b g — falls out of the dW tiles of k-block 0 (the A operand IS g)
// One wave = one 32x32 output tile; tiles [0, tilesW) are dW, the rest dx.
// ---------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void fc_bwd_kernel(const float *__restrict__ g, const float *__restrict__ x,
                                                     const float *__restrict__ W, int B, int K, int Nout,
                                                     int tilesW, int tilesX, float *__restrict__ dW,
                                                     float *__restrict__ dx, float *__restrict__ db, int zero_db) {
    // One WORKGROUP = one 32x32 output tile; its four waves each contract a quarter of the reduction range and meet in
    // LDS (a tile's contraction is a chain of up to 512 dependent MFMAs: one wave per tile left the launch 4x off the
    // MFMA time with 1.5 workgroups per CU; split four ways there are 6 balanced workgroups per CU).
    __shared__ float red[3 * 16 * 64];
    __shared__ float dred[3 * 32];
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = blockIdx.x;
    const int kblocks = (K + 31) >> 5;
    f32x16 acc = {0};
    const bool is_w = wid < tilesW;
    int orow0 = 0, ocol = 0;          // output tile: rows orow0 + mfma_row, column ocol (per lane)
    float dbs = 0.f;
    if (is_w) {
        const int nb = wid / kblocks, kbk = wid - nb * kblocks;
        const int n = nb * 32 + j, kc = kbk * 32 + j;
        const bool nv = n < Nout, kv = kc < K;
        const float *gp = g + (nv ? n : 0), *xp = x + (kv ? kc : 0);
        const int per = (((B + 3) / 4) + 15) & ~15;               // rows of the batch per wave, a multiple of 16
        const int bb = wave * per, be = (bb + per < B) ? bb + per : B;
        float av[8], bv[8];
        auto fetch = [&](int b0, float (&a)[8], float (&bq)[8]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 2 * u + h;
                const bool ok = b < be;
                a[u] = (ok && nv) ? gp[(size_t)b * Nout] : 0.f;
                bq[u] = (ok && kv) ? xp[(size_t)b * K] : 0.f;
            }
        };
        if (bb < be) fetch(bb, av, bv);
        for (int b0 = bb; b0 < be; b0 += 16) {   // group i+1's 16 loads in flight while group i's 8 MFMAs issue
            float na[8], nbv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { na[u] = 0.f; nbv[u] = 0.f; }
            if (b0 + 16 < be) fetch(b0 + 16, na, nbv);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc = mfma32(av[u], bv[u], acc); dbs += av[u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { av[u] = na[u]; bv[u] = nbv[u]; }
        }
        orow0 = nb * 32; ocol = kc;
        dbs += __shfl_xor(dbs, 32);
    } else if (dx) {
        const int t = wid - tilesW;
        const int rb = t / kblocks, kbk = t - rb * kblocks;
        int row = rb * 32 + j; row = row < B ? row : B - 1;
        const int kc = kbk * 32 + j;
        const bool kv = kc < K;
        const float *wp = W + (kv ? kc : 0);
        const float *gr = g + (size_t)row * Nout;
        if (VEC) {   // Nout % 8 == 0: k-block = 8 values of n, lane (j,h) holds n = 8kb + 4h .. +3
            const f32x4 *ap = (const f32x4 *)gr + h;
            const int KB = Nout >> 3;
            const int per = (((KB + 3) / 4) + 1) & ~1;            // k-blocks per wave, even
            const int k0 = wave * per, k1 = (k0 + per < KB) ? k0 + per : KB;
            f32x4 a[2]; float w[2][4];
            auto fetch = [&](int kb, f32x4 (&aa)[2], float (&ww)[2][4]) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bool ok = kb + u < k1;
                    const int kk = ok ? kb + u : (k1 > 0 ? k1 - 1 : 0);
                    aa[u] = ok ? ap[kk * 2] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) ww[u][e] = (kv && ok) ? wp[(size_t)(kk * 8 + 4 * h + e) * K] : 0.f;
                }
            };
            if (k0 < k1) fetch(k0, a, w);
            for (int kb = k0; kb < k1; kb += 2) {
                f32x4 na[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                float nw[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                if (kb + 2 < k1) fetch(kb + 2, na, nw);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = mfma32(a[u][e], w[u][e], acc);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    a[u] = na[u];
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[u][e] = nw[u][e];
                }
            }
        } else {
            const int per = (((Nout + 3) / 4) + 7) & ~7;          // values of n per wave, a multiple of 8
            const int n0w = wave * per, n1w = (n0w + per < Nout) ? n0w + per : Nout;
            for (int n0 = n0w; n0 < n1w; n0 += 8) {
                float av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int n = n0 + 2 * u + h;
                    const bool ok = n < n1w;
                    av[u] = ok ? gr[n] : 0.f;
                    bv[u] = (ok && kv) ? wp[(size_t)n * K] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = mfma32(av[u], bv[u], acc);
            }
        }
        orow0 = rb * 32; ocol = kc;
    }
    // meet in LDS: waves 1..3 park their tile, wave 0 adds and stores
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
        if (h == 0) dred[(wave - 1) * 32 + j] = dbs;
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
    if (is_w) {
        const int kbk = wid % kblocks;
        if (ocol < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = orow0 + mfma_row(r, lane);
                if (row < Nout) dW[(size_t)row * K + ocol] = acc[r];
            }
        }
        const int n = orow0 + j;
        // zero_db: the layer feeds a train-mode BatchNorm, whose backward makes sum_b g exactly zero in exact arithmetic
        // (the conv biases of the trunks are treated the same way): write the exact value, not its rounding residue
        if (kbk == 0 && h == 0 && n < Nout) db[n] = zero_db ? 0.f : dbs + dred[j] + dred[32 + j] + dred[64 + j];
    } else if (dx) {
        if (ocol < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orow = orow0 + mfma_row(r, lane);
                if (orow < B) dx[(size_t)orow * K + ocol] = acc[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
static TrainChan make_chan(const float *w1, const float *b1, const float *s1c, const float *t1c,
                           const float *w2p, const float *s2c, const float *t2c, const void *w2x = nullptr) {
    TrainChan P; P.w1 = w1; P.b1 = b1; P.s1c = s1c; P.t1c = t1c; P.w2p = w2p; P.s2c = s2c; P.t2c = t2c;
    P.w2x = (const u16 *)w2x;
    return P;
}

// nterms: 0 = fp32 (w2p), 1 / 3 = bf16 / bf16x3 (w2x).  One implementation behind the fp32 and the _bf entry points.
static int bn2_stats_impl(const float *x, int B, int N, const float *trans, const float *w1, const float *b1,
                          const float *s1c, const float *t1c, const float *w2p, const void *w2x, int nterms, int S,
                          float *part, float *z2t, void *stream) {
    if (!x || !w1 || !b1 || !s1c || !t1c || !(nterms ? w2x : (const void *)w2p) || !part || B <= 0 || N <= 0 ||
        (nterms != 0 && nterms != 1 && nterms != 3))
        return PNGPD_ERR_INVALID_ARG;
    const int T = (N + TP - 1) / TP;
    if (!splits_ok(S, T)) return PNGPD_ERR_INVALID_ARG;
    TrainChan P = make_chan(w1, b1, s1c, t1c, w2p, nullptr, nullptr, w2x);
    const size_t lds = (TP * H1S + 3 * TP) * sizeof(float);
    const dim3 grid((unsigned)B * S);
    hipStream_t sm = (hipStream_t)stream;
    if (nterms == 0)
        hipLaunchKernelGGL(trunk_bn2_stats_kernel<0>, grid, dim3(256), lds, sm, x, N, trans, P, T, S, part, (f32x4 *)z2t);
    else if (nterms == 1)
        hipLaunchKernelGGL(trunk_bn2_stats_kernel<1>, grid, dim3(256), lds, sm, x, N, trans, P, T, S, part, (f32x4 *)z2t);
    else
        hipLaunchKernelGGL(trunk_bn2_stats_kernel<3>, grid, dim3(256), lds, sm, x, N, trans, P, T, S, part, (f32x4 *)z2t);
    return pngpd_launch_status();
}

int pngpd_bwd_gather_impl(const float *x, int B, int N, const float *trans, const float *w1, const float *b1,
                           const float *s1c, const float *t1c, const float *w2p, const void *w2x, int nterms,
                           const float *s2c, const float *t2c, const int *idx, const float *coef,
                           int clouds_per_range, float *Gp, const ACvecArgs *tail, void *stream) {
    if (!x || !w1 || !b1 || !s1c || !t1c || !(nterms ? w2x : (const void *)w2p) || !s2c || !t2c || !idx || !coef ||
        !Gp || B <= 0 || N <= 0 || clouds_per_range <= 0 || (nterms != 0 && nterms != 1 && nterms != 3))
        return PNGPD_ERR_INVALID_ARG;
    const int R = (B + clouds_per_range - 1) / clouds_per_range;
    TrainChan P = make_chan(w1, b1, s1c, t1c, w2p, s2c, t2c, w2x);
    const size_t lds = (TP * H1S + 8 * TP) * sizeof(float);
    const void *fn = nterms == 0 ? (const void *)trunk_bwd_gather_kernel<0>
                   : nterms == 1 ? (const void *)trunk_bwd_gather_kernel<1> : (const void *)trunk_bwd_gather_kernel<3>;
    int st = pngpd_allow_lds(fn, lds);
    if (st != PNGPD_OK) return st;
    const int n_main = R * 16;
    const dim3 grid((unsigned)(n_main + (tail ? 128 : 0)));   // + one workgroup per row of A
    const ACvecArgs AT = tail ? *tail : ACvecArgs{};
    static_assert((TP * H1S + 8 * TP) * sizeof(float) >= ACVEC_LDS_DOUBLES * sizeof(double), "tail LDS");
    hipStream_t sm = (hipStream_t)stream;
    if (nterms == 0)
        hipLaunchKernelGGL(trunk_bwd_gather_kernel<0>, grid, dim3(256), lds, sm, x, B, N, trans, P, idx, coef, clouds_per_range, Gp, AT, n_main);
    else if (nterms == 1)
        hipLaunchKernelGGL(trunk_bwd_gather_kernel<1>, grid, dim3(256), lds, sm, x, B, N, trans, P, idx, coef, clouds_per_range, Gp, AT, n_main);
    else
        hipLaunchKernelGGL(trunk_bwd_gather_kernel<3>, grid, dim3(256), lds, sm, x, B, N, trans, P, idx, coef, clouds_per_range, Gp, AT, n_main);
    return pngpd_launch_status();
}

template <bool LOADZ, int NT>
static int launch_bwd_d(dim3 grid, size_t lds, hipStream_t sm, const float *x, int N, const float *trans,
                        const TrainChan &P, const BwdDParams &D, int T, int S, const float *z2t, float *g2t,
                        float *pa, float *ps2) {
    int st = pngpd_allow_lds((const void *)trunk_bwd_d_kernel<LOADZ, NT>, lds);
    if (st != PNGPD_OK) return st;
    hipLaunchKernelGGL((trunk_bwd_d_kernel<LOADZ, NT>), grid, dim3(256), lds, sm,
                       x, N, trans, P, D, T, S, (const f32x4 *)z2t, (f32x4 *)g2t, pa, ps2);
    return pngpd_launch_status();
}

template <bool LOADZ, int NT>
static int launch_bwd_e(dim3 grid, size_t lds, hipStream_t sm, const float *x, int N, const float *trans,
                        const TrainChan &P, const BwdEParams &E, int T, int S, const float *z2t, const float *g2t,
                        float *pc, float *pR, float *pW2, const DW3Args *tail) {
    int st = pngpd_allow_lds((const void *)trunk_bwd_e_kernel<LOADZ, NT>, lds);
    if (st != PNGPD_OK) return st;
    static_assert((BWD_E_LDS_FLOATS + K128_LDS_FLOATS) * sizeof(float) >= DW3_LDS_DOUBLES * sizeof(double), "tail LDS");
    const int n_main = (int)grid.x;
    if (tail) grid.x += 1024 / DW3_CPB;
    hipLaunchKernelGGL((trunk_bwd_e_kernel<LOADZ, NT>), grid, dim3(256), lds, sm,
                       x, N, trans, P, E, T, S, (const f32x4 *)z2t, (const f32x4 *)g2t, pc, pR, pW2,
                       tail ? *tail : DW3Args{}, n_main);
    return pngpd_launch_status();
}

extern "C" {

// Workgroups per cloud for the training passes when the caller names no target (target_blocks <= 0).  512 workgroups are
// resident (2 per CU); every workgroup leaves ~80 KB of partial sums (Gram blocks, dW2 share) that a reduce launch
// streams back, so at small batches "one tile per workgroup" both overshoots a resident round and triples the partial
// traffic.  Cost of S, in tile times, fitted to tools/bench_train_splits.py (B 32..256, N 750 / 1024, and B = N = 1024):
//   rounds(B S / 512) * (ceil(T/S) * (B S <= 256 ? 0.6 : 1) + 0.5)  +  0.5 * B S / 512
// (0.6: a workgroup alone on its CU does not share the MFMA pipe; 0.5: prologue; last term: partial write + reduce).
// Measured against the old rule (aim at 1024 workgroups): B=64 N=750 0.960 -> 0.853 ms per step, B=128 N=750 1.483 ->
// 1.189 ms, B=256 N=1024 2.400 -> 2.319 ms, B=64 N=1024 1.018 -> 0.945 ms; B >= 512 unchanged (S = 1).
static int train_auto_splits(int B, int T) {
    int bestS = 1;
    double best = 1e300;
    for (int S = 1; S <= T; ++S) {
        const long G = (long)B * S;
        const double tiles = (double)((T + S - 1) / S) * (G <= 256 ? 0.6 : 1.0);
        const double v = (double)((G + 511) / 512) * (tiles + 0.5) + 0.5 * (double)G / 512.0;
        if (v < best - 1e-9) { best = v; bestS = S; }
    }
    return bestS;
}

int pngpd_trunk_splits(int B, int N, int target_blocks) {
    if (B <= 0 || N <= 0) return 0;
    const int T = (N + TP - 1) / TP;
    return target_blocks > 0 ? pngpd_splits_for(B, T, target_blocks) : train_auto_splits(B, T);
}

size_t pngpd_trunk_g2t_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return (size_t)B * ((N + TP - 1) / TP) * TP * 128 * sizeof(float);
}

int pngpd_cloud_moments(const float *x, int B, int N, double *mom, void *stream) {
    if (!x || !mom || B <= 0 || N <= 0) return PNGPD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(cloud_moments_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, N, mom);
    return pngpd_launch_status();
}

int pngpd_trunk_bn2_stats(const float *x, int B, int N, const float *trans,
                          const float *w1, const float *b1, const float *s1c, const float *t1c,
                          const float *w2p, int S, float *part, float *z2t, void *stream) {
    return bn2_stats_impl(x, B, N, trans, w1, b1, s1c, t1c, w2p, nullptr, 0, S, part, z2t, stream);
}

int pngpd_trunk_bn2_stats_bf(const float *x, int B, int N, const float *trans,
                             const float *w1, const float *b1, const float *s1c, const float *t1c,
                             const void *w2x, int nterms, int S, float *part, void *z2t, void *stream) {
    if (nterms != 1 && nterms != 3) return PNGPD_ERR_INVALID_ARG;
    return bn2_stats_impl(x, B, N, trans, w1, b1, s1c, t1c, nullptr, w2x, nterms, S, part, (float *)z2t, stream);
}

int pngpd_trunk_fwd_train(const float *x, int B, int N, const float *trans,
                          const float *w1, const float *b1, const float *s1c, const float *t1c,
                          const float *w2p, const float *s2c, const float *t2c, const float *w3sp, int S,
                          float *pmax, int *parg, float *psum, float *psh, const float *z2t, void *stream) {
    if (!x || !w1 || !b1 || !s1c || !t1c || !w2p || !s2c || !t2c || !w3sp || !pmax || !parg || !psum || !psh ||
        B <= 0 || N <= 0)
        return PNGPD_ERR_INVALID_ARG;
    const int T = (N + TP - 1) / TP;
    if (!splits_ok(S, T)) return PNGPD_ERR_INVALID_ARG;
    TrainChan P = make_chan(w1, b1, s1c, t1c, w2p, s2c, t2c);
    const size_t lds = TRAIN_MAIN_LDS_FLOATS * sizeof(float);
    int st = pngpd_allow_lds(z2t ? (const void *)trunk_fwd_train_kernel<true> : (const void *)trunk_fwd_train_kernel<false>, lds);
    if (st != PNGPD_OK) return st;
    if (z2t)
        hipLaunchKernelGGL(trunk_fwd_train_kernel<true>, dim3((unsigned)B * S), dim3(256), lds, (hipStream_t)stream,
                           x, N, trans, P, w3sp, T, S, pmax, parg, psum, psh, (const f32x4 *)z2t);
    else
        hipLaunchKernelGGL(trunk_fwd_train_kernel<false>, dim3((unsigned)B * S), dim3(256), lds, (hipStream_t)stream,
                           x, N, trans, P, w3sp, T, S, pmax, parg, psum, psh, (const f32x4 *)nullptr);
    return pngpd_launch_status();
}

int pngpd_trunk_bwd_gather(const float *x, int B, int N, const float *trans,
                           const float *w1, const float *b1, const float *s1c, const float *t1c,
                           const float *w2p, const float *s2c, const float *t2c,
                           const int *idx, const float *coef, int clouds_per_range, float *Gp, void *stream) {
    return pngpd_bwd_gather_impl(x, B, N, trans, w1, b1, s1c, t1c, w2p, nullptr, 0, s2c, t2c, idx, coef, clouds_per_range,
                           Gp, nullptr, stream);
}

int pngpd_trunk_bwd_gather_bf(const float *x, int B, int N, const float *trans,
                              const float *w1, const float *b1, const float *s1c, const float *t1c,
                              const void *w2x, int nterms, const float *s2c, const float *t2c,
                              const int *idx, const float *coef, int clouds_per_range, float *Gp, void *stream) {
    if (nterms != 1 && nterms != 3) return PNGPD_ERR_INVALID_ARG;
    return pngpd_bwd_gather_impl(x, B, N, trans, w1, b1, s1c, t1c, nullptr, w2x, nterms, s2c, t2c, idx, coef,
                           clouds_per_range, Gp, nullptr, stream);
}

int pngpd_trunk_pool_refine(const float *x, int B, int N, const float *trans,
                            const float *w1, const float *b1, const float *s1c, const float *t1c,
                            const float *w2p, const float *s2c, const float *t2c,
                            const float *w3sp, const float *w3, const float *g3, const int *idx,
                            int clouds_per_range, int variant, float *zex, void *stream) {
    if (!x || !w1 || !b1 || (!s1c != !t1c) || !w2p || !s2c || !t2c || !idx || !zex || B <= 0 || N <= 0 ||
        clouds_per_range <= 0 || variant < 0 || variant > 3 ||
        (variant == 0 ? !w3sp : ((variant == 3 || !w3sp) && (!w3 || !g3))))
        return PNGPD_ERR_INVALID_ARG;
    const int R = (B + clouds_per_range - 1) / clouds_per_range;
    TrainChan P = make_chan(w1, b1, s1c, t1c, w2p, s2c, t2c);
    if ((variant == 1 || variant == 2) && w3sp && N <= 65536) {
        // the VALU variants with the sign-folded MFMA_B weights at hand: layers 1-2 at the DISTINCT arg-max points only
        const int W32 = (N + 31) / 32;
        const size_t lds = (size_t)REFINE_DEDUP_LDS_FLOATS(W32) * sizeof(float);
        const void *fn = variant == 1 ? (const void *)trunk_pool_refine_dedup_kernel<1>
                                      : (const void *)trunk_pool_refine_dedup_kernel<2>;
        int st = pngpd_allow_lds(fn, lds);
        if (st != PNGPD_OK) return st;
        const dim3 grid((unsigned)(B < 2048 ? B : 2048));
        if (variant == 1)
            hipLaunchKernelGGL(trunk_pool_refine_dedup_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, x, B, N, trans,
                               P, w3sp, idx, W32, zex);
        else
            hipLaunchKernelGGL(trunk_pool_refine_dedup_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, x, B, N, trans,
                               P, w3sp, idx, W32, zex);
        return pngpd_launch_status();
    }
    if (variant != 0 && (!w3 || !g3)) return PNGPD_ERR_INVALID_ARG;
    const size_t lds = REFINE_LDS_FLOATS * sizeof(float);
    const void *fn = variant == 0 ? (const void *)trunk_pool_refine_kernel<0>
                   : variant == 1 ? (const void *)trunk_pool_refine_kernel<1>
                   : variant == 2 ? (const void *)trunk_pool_refine_kernel<2> : (const void *)trunk_pool_refine_kernel<3>;
    int st = pngpd_allow_lds(fn, lds);
    if (st != PNGPD_OK) return st;
    const dim3 grid((unsigned)R * 16);
    hipStream_t sm = (hipStream_t)stream;
#define PNGPD_REFINE_LAUNCH(V) hipLaunchKernelGGL(trunk_pool_refine_kernel<V>, grid, dim3(256), lds, sm, x, B, N, trans, \
                                                  P, w3sp, w3, g3, idx, clouds_per_range, zex)
    if (variant == 0) PNGPD_REFINE_LAUNCH(0);
    else if (variant == 1) PNGPD_REFINE_LAUNCH(1);
    else if (variant == 2) PNGPD_REFINE_LAUNCH(2);
    else PNGPD_REFINE_LAUNCH(3);
#undef PNGPD_REFINE_LAUNCH
    return pngpd_launch_status();
}

int pngpd_trunk_bwd_d(const float *x, int B, int N, const float *trans,
                      const float *w1, const float *b1, const float *s1c, const float *t1c,
                      const float *w2p, const float *s2c, const float *t2c,
                      const float *is2, const float *nm2, const float *Ap, const float *cvec,
                      const float *w3, const int *idx, const float *coef, const float *z2t, int S,
                      float *g2t, float *pa, float *ps2, void *stream) {
    if (!x || !w1 || !b1 || !s1c || !t1c || !w2p || !s2c || !t2c || !is2 || !nm2 || !Ap || !cvec || !w3 ||
        !idx || !coef || !g2t || !pa || !ps2 || B <= 0 || N <= 0)
        return PNGPD_ERR_INVALID_ARG;
    const int T = (N + TP - 1) / TP;
    if (!splits_ok(S, T)) return PNGPD_ERR_INVALID_ARG;
    if (N > (1 << 30)) return PNGPD_ERR_UNSUPPORTED;
    TrainChan P = make_chan(w1, b1, s1c, t1c, w2p, s2c, t2c);
    BwdDParams D; D.is2 = is2; D.nm2 = nm2; D.Ap = Ap; D.Ax = nullptr; D.cvec = cvec; D.w3 = w3; D.idx = idx; D.coef = coef;
    const size_t lds = (z2t ? BWD_D_LDS_FLOATS - TP * H1S + BWD_D_APL_FLOATS : BWD_D_LDS_FLOATS) * sizeof(float);
    const dim3 grid((unsigned)B * S);
    return z2t ? launch_bwd_d<true, 0>(grid, lds, (hipStream_t)stream, x, N, trans, P, D, T, S, z2t, g2t, pa, ps2)
               : launch_bwd_d<false, 0>(grid, lds, (hipStream_t)stream, x, N, trans, P, D, T, S, nullptr, g2t, pa, ps2);
}

int pngpd_trunk_bwd_d_bf(const float *x, int B, int N, const float *s2c, const float *t2c,
                         const float *is2, const float *nm2, const void *Ax, int nterms, const float *cvec,
                         const float *w3, const int *idx, const float *coef, const void *z2tv, int S,
                         void *g2tv, float *pa, float *ps2, void *stream) {
    const float *z2t = (const float *)z2tv;
    float *g2t = (float *)g2tv;
    if (!x || !s2c || !t2c || !is2 || !nm2 || !Ax || !cvec || !w3 || !idx || !coef || !z2t || !g2t || !pa || !ps2 ||
        B <= 0 || N <= 0 || (nterms != 1 && nterms != 3))
        return PNGPD_ERR_INVALID_ARG;
    const int T = (N + TP - 1) / TP;
    if (!splits_ok(S, T)) return PNGPD_ERR_INVALID_ARG;
    if (N > (1 << 30)) return PNGPD_ERR_UNSUPPORTED;
    TrainChan P = make_chan(nullptr, nullptr, nullptr, nullptr, nullptr, s2c, t2c);   // z2 is read back: layers 1-2 unused
    BwdDParams D; D.is2 = is2; D.nm2 = nm2; D.Ap = nullptr; D.Ax = (const u16 *)Ax; D.cvec = cvec; D.w3 = w3;
    D.idx = idx; D.coef = coef;
    const dim3 grid((unsigned)B * S);
#ifdef PNGPD_BF_LEGACY_D      // the inherited kernel (fp32 LDS tiles, operands converted at every read): A/B builds only
    const size_t lds = (BWD_D_LDS_FLOATS - TP * H1S) * sizeof(float);
    return nterms == 1 ? launch_bwd_d<true, 1>(grid, lds, (hipStream_t)stream, x, N, nullptr, P, D, T, S, z2t, g2t, pa, ps2)
                       : launch_bwd_d<true, 3>(grid, lds, (hipStream_t)stream, x, N, nullptr, P, D, T, S, z2t, g2t, pa, ps2);
#else
    const size_t lds = nterms == 1 ? DBF_LDS_BYTES(1) : DBF_LDS_BYTES(3);
    const void *fn = nterms == 1 ? (const void *)trunk_bwd_d_bf_kernel<1> : (const void *)trunk_bwd_d_bf_kernel<3>;
    int st = pngpd_allow_lds(fn, lds);
    if (st != PNGPD_OK) return st;
    if (nterms == 1)
        hipLaunchKernelGGL(trunk_bwd_d_bf_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, N, P, D, T, S,
                           (const f32x4 *)z2t, (f32x4 *)g2t, pa, ps2);
    else
        hipLaunchKernelGGL(trunk_bwd_d_bf_kernel<3>, grid, dim3(256), lds, (hipStream_t)stream, N, P, D, T, S,
                           (const f32x4 *)z2t, (f32x4 *)g2t, pa, ps2);
    return pngpd_launch_status();
#endif
}

int pngpd_trunk_bwd_e(const float *x, int B, int N, const float *trans,
                      const float *w1, const float *b1, const float *s1c, const float *t1c,
                      const float *w2p, const float *is1, const float *nm1, const float *is2, const float *nm2,
                      const float *a1m, const float *a2m, const float *dsc2, const float *w2tp,
                      const float *z2t, const float *g2t, int S, float *pc, float *pR, float *pW2, void *stream) {
    if (!w2p || !w2tp) return PNGPD_ERR_INVALID_ARG;
    return pngpd_bwd_e_impl(x, B, N, trans, w1, b1, s1c, t1c, w2p, is1, nm1, is2, nm2, a1m, a2m, dsc2, w2tp, nullptr, 0,
                            z2t, g2t, S, pc, pR, pW2, nullptr, stream);
}

int pngpd_trunk_bwd_e_bf(const float *x, int B, int N, const float *trans,
                         const float *w1, const float *b1, const float *s1c, const float *t1c,
                         const float *is1, const float *nm1, const float *is2, const float *nm2,
                         const float *a1m, const float *a2m, const float *dsc2, const void *w2tx, int nterms,
                         const void *z2tv, const void *g2tv, int S, float *pc, float *pR, float *pW2, void *stream) {
    if (!w2tx || !z2tv || (nterms != 1 && nterms != 3)) return PNGPD_ERR_INVALID_ARG;
    return pngpd_bwd_e_impl(x, B, N, trans, w1, b1, s1c, t1c, nullptr, is1, nm1, is2, nm2, a1m, a2m, dsc2, nullptr, w2tx,
                            nterms, (const float *)z2tv, (const float *)g2tv, S, pc, pR, pW2, nullptr, stream);
}

}  // extern "C"

// Pass E behind both C entries (nterms 0: fp32, w2p / w2tp; 1 / 3: bf16 / bf16x3, w2tx, z2t required), optionally with
// dW3's finalize as tail workgroups (tail != NULL: the fused backward, pngpd_train_step.hip).
int pngpd_bwd_e_impl(const float *x, int B, int N, const float *trans, const float *w1, const float *b1,
                     const float *s1c, const float *t1c, const float *w2p, const float *is1, const float *nm1,
                     const float *is2, const float *nm2, const float *a1m, const float *a2m, const float *dsc2,
                     const float *w2tp, const void *w2tx, int nterms, const float *z2t, const float *g2t, int S,
                     float *pc, float *pR, float *pW2, const DW3Args *tail, void *stream) {
    if (!x || !w1 || !b1 || !s1c || !t1c || !is1 || !nm1 || !is2 || !nm2 || !a1m || !a2m || !dsc2 || !g2t || !pc ||
        !pR || !pW2 || B <= 0 || N <= 0 || (nterms != 0 && nterms != 1 && nterms != 3))
        return PNGPD_ERR_INVALID_ARG;
    if (nterms ? (!w2tx || !z2t) : (!w2p || !w2tp)) return PNGPD_ERR_INVALID_ARG;
    const int T = (N + TP - 1) / TP;
    if (!splits_ok(S, T)) return PNGPD_ERR_INVALID_ARG;
    TrainChan P = make_chan(w1, b1, s1c, t1c, w2p, nullptr, nullptr);
    BwdEParams E; E.is1 = is1; E.nm1 = nm1; E.is2 = is2; E.nm2 = nm2; E.a1m = a1m; E.a2m = a2m; E.dsc2 = dsc2;
    E.w2tp = w2tp; E.w2tx = (const u16 *)w2tx;
    const size_t lds = (BWD_E_LDS_FLOATS + K128_LDS_FLOATS) * sizeof(float);   // 79 KB: two workgroups per CU still fit
    const dim3 grid((unsigned)B * S);
    hipStream_t sm = (hipStream_t)stream;
#ifdef PNGPD_BF_LEGACY_E     // the inherited kernel (fp32 LDS tiles, operands converted at every read): A/B builds only
    if (nterms == 1) return launch_bwd_e<true, 1>(grid, lds, sm, x, N, trans, P, E, T, S, z2t, g2t, pc, pR, pW2, tail);
    if (nterms == 3) return launch_bwd_e<true, 3>(grid, lds, sm, x, N, trans, P, E, T, S, z2t, g2t, pc, pR, pW2, tail);
#else
    if (nterms) {
        const size_t ldsb = nterms == 1 ? EBF_LDS_BYTES(1) : EBF_LDS_BYTES(3);
        static_assert(EBF_LDS_BYTES(1) >= DW3_LDS_DOUBLES * sizeof(double) && EBF_LDS_BYTES(3) <= 81920, "pass E (bf) LDS");
        const void *fn = nterms == 1 ? (const void *)trunk_bwd_e_bf_kernel<1> : (const void *)trunk_bwd_e_bf_kernel<3>;
        int st = pngpd_allow_lds(fn, ldsb);
        if (st != PNGPD_OK) return st;
        dim3 g = grid;
        const int n_main = (int)g.x;
        if (tail) g.x += 1024 / DW3_CPB;
        const DW3Args WT = tail ? *tail : DW3Args{};
        if (nterms == 1)
            hipLaunchKernelGGL(trunk_bwd_e_bf_kernel<1>, g, dim3(256), ldsb, sm, x, N, trans, P, E, T, S, (const f32x4 *)z2t,
                               (const f32x4 *)g2t, pc, pR, pW2, WT, n_main);
        else
            hipLaunchKernelGGL(trunk_bwd_e_bf_kernel<3>, g, dim3(256), ldsb, sm, x, N, trans, P, E, T, S, (const f32x4 *)z2t,
                               (const f32x4 *)g2t, pc, pR, pW2, WT, n_main);
        return pngpd_launch_status();
    }
#endif
    return z2t ? launch_bwd_e<true, 0>(grid, lds, sm, x, N, trans, P, E, T, S, z2t, g2t, pc, pR, pW2, tail)
               : launch_bwd_e<false, 0>(grid, lds, sm, x, N, trans, P, E, T, S, nullptr, g2t, pc, pR, pW2, tail);
}

int pngpd_fc_bwd_impl(const float *g, const float *x, const float *W, int B, int K, int Nout,
                      float *dW, float *dx, float *db, int zero_db, void *stream) {
    if (!g || !x || !W || !dW || !db || B <= 0 || K <= 0 || Nout <= 0) return PNGPD_ERR_INVALID_ARG;
    const int kblocks = (K + 31) / 32;
    const int tilesW = ((Nout + 31) / 32) * kblocks, tilesX = dx ? ((B + 31) / 32) * kblocks : 0;
    const unsigned grid = (unsigned)(tilesW + tilesX);   // one workgroup per output tile (its waves split the contraction)
    if ((Nout & 7) == 0)
        hipLaunchKernelGGL(fc_bwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, x, W, B, K, Nout,
                           tilesW, tilesX, dW, dx, db, zero_db);
    else
        hipLaunchKernelGGL(fc_bwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, x, W, B, K, Nout,
                           tilesW, tilesX, dW, dx, db, zero_db);
    return pngpd_launch_status();
}

extern "C" {

int pngpd_fc_bwd(const float *g, const float *x, const float *W, int B, int K, int Nout,
                 float *dW, float *dx, float *db, void *stream) {
    return pngpd_fc_bwd_impl(g, x, W, B, K, Nout, dW, dx, db, 0, stream);
}

int pngpd_bn1d_fwd_train(const float *z, int B, int C, const float *gamma, const float *beta, float eps,
                         int relu, float *y, float *mean, float *var, float momentum, float *rm, float *rv,
                         long long *nbt, void *stream) {
    if (!z || !gamma || !beta || !y || !mean || !var || B <= 0 || C <= 0 || (rm && !rv)) return PNGPD_ERR_INVALID_ARG;
    const dim3 grid((C + BN1D_CW - 1) / BN1D_CW);
    if (B <= BN1D_RL * BN1D_NV)
        hipLaunchKernelGGL(bn1d_fwd_train_kernel<true>, grid, dim3(BN1D_CW * BN1D_RL), 0, (hipStream_t)stream,
                           z, B, C, gamma, beta, eps, relu, y, mean, var, momentum, rm, rv, nbt);
    else
        hipLaunchKernelGGL(bn1d_fwd_train_kernel<false>, grid, dim3(BN1D_CW * BN1D_RL), 0, (hipStream_t)stream,
                           z, B, C, gamma, beta, eps, relu, y, mean, var, momentum, rm, rv, nbt);
    return pngpd_launch_status();
}

int pngpd_bn1d_bwd(const float *dy, const float *z, const float *y, int B, int C, const float *gamma,
                   const float *mean, const float *var, float eps, int relu,
                   float *dz, float *dgamma, float *dbeta, void *stream) {
    if (!dy || !z || !y || !gamma || !mean || !var || !dz || !dgamma || !dbeta || B <= 0 || C <= 0)
        return PNGPD_ERR_INVALID_ARG;
    const dim3 grid((C + BN1D_CW - 1) / BN1D_CW);
    if (B <= BN1D_RL * BN1D_NV)
        hipLaunchKernelGGL(bn1d_bwd_kernel<true>, grid, dim3(BN1D_CW * BN1D_RL), 0, (hipStream_t)stream,
                           dy, z, y, B, C, gamma, mean, var, eps, relu, dz, dgamma, dbeta);
    else
        hipLaunchKernelGGL(bn1d_bwd_kernel<false>, grid, dim3(BN1D_CW * BN1D_RL), 0, (hipStream_t)stream,
                           dy, z, y, B, C, gamma, mean, var, eps, relu, dz, dgamma, dbeta);
    return pngpd_launch_status();
}

int pngpd_nll_fwd(const float *logp, const long long *target, int B, int K, int mean, float *loss, void *stream) {
    if (!logp || !target || !loss || B <= 0 || K <= 0) return PNGPD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(nll_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logp, target, B, K, mean, loss);
    return pngpd_launch_status();
}

int pngpd_nll_log_softmax_bwd(const float *g, const float *gloss, const long long *target, const float *logp, int B,
                              int K, int mean, float *dlogits, void *stream) {
    if (!gloss || !target || !logp || !dlogits || B <= 0 || K <= 0) return PNGPD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(nll_log_softmax_bwd_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, gloss,
                       target, logp, B, K, mean, dlogits);
    return pngpd_launch_status();
}

int pngpd_log_softmax_bwd(const float *g, const float *logp, int B, int K, float *dlogits, void *stream) {
    if (!g || !logp || !dlogits || B <= 0 || K <= 0) return PNGPD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       g, logp, B, K, dlogits);
    return pngpd_launch_status();
}

}  // extern "C"
