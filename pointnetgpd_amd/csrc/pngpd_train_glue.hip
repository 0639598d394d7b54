// libpngpd — parameter-sized fp64 algebra between the training passes ("finalize" kernels):
// BatchNorm statistics -> per-channel affine forms (+ running-stat update), pooled output from the
// per-workgroup maxima, deterministic fp64 reduction of per-workgroup partial sums, and the closed-form
// weight gradients / pass-D operands (DESIGN.md "Training passes"; algebra verified in fp64 against
// autograd by tests/train_algo_prototype.py).  Everything here is tiny (<= 1024x128 matrices); the point is
// to replace ~500 launch-bound framework ops per step by ~40 kernels on the same stream.
#include "pngpd_common.h"
#include "pngpd_internal.h"
#include "pngpd_glue_bodies.h"

#define NT 256

__device__ __forceinline__ double block_sum(double v, double *red) {   // red: [>= NT/64] shared; fixed tree: deterministic
    // wave tree by shuffles, then the NT/64 wave partials through LDS: 2 barriers (an LDS tree costs 10 per call, and
    // bn1_finalize makes 12 calls)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) r += red[w];
    __syncthreads();
    return r;
}

__device__ __forceinline__ void running_update(float *rm, float *rv, int c, double mean, double var_b,
                                               double M, double mom) {
    if (!rm) return;
    const double unb = var_b * (M / (M > 1.0 ? M - 1.0 : 1.0));
    rm[c] = (float)((1.0 - mom) * (double)rm[c] + mom * mean);
    rv[c] = (float)((1.0 - mom) * (double)rv[c] + mom * unb);
}

// ---------------------------------------------------------------------------------------
// BN1: closed-form batch statistics of z1 = W1 x' + b1 from the per-cloud input moments.
//   mom (B,9) f64 = {sx,sy,sz,sxx,sxy,sxz,syy,syz,szz};  trans (B,3,3) or NULL (x' = x^T T)
//   chan (4,64) f32 = s1c, t1c, is1, nm1 ;  stats f64 = mx[3], Cx[9], mu1[64], var1[64]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void bn1_finalize_kernel(
    const double *__restrict__ mom, const float *__restrict__ trans, int B, double M,
    const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ g1,
    const float *__restrict__ be1, double eps, double momentum, float *rm, float *rv, long long *nbt,
    float *__restrict__ chan, double *__restrict__ stats) {
    __shared__ double red[NT];
    __shared__ double tot[12];
    const int tid = threadIdx.x;
    double a[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) a[i] = 0.0;
    for (int b = tid; b < B; b += NT) {
        const double *m = mom + (size_t)b * 9;
        const double mb[3] = {m[0], m[1], m[2]};
        const double S[3][3] = {{m[3], m[4], m[5]}, {m[4], m[6], m[7]}, {m[5], m[7], m[8]}};
        if (trans) {
            double T[3][3];
#pragma unroll
            for (int i = 0; i < 9; ++i) T[i / 3][i % 3] = (double)trans[(size_t)b * 9 + i];
#pragma unroll
            for (int j = 0; j < 3; ++j) a[j] += mb[0] * T[0][j] + mb[1] * T[1][j] + mb[2] * T[2][j];
            double ST[3][3];   // S T
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) ST[i][j] = S[i][0] * T[0][j] + S[i][1] * T[1][j] + S[i][2] * T[2][j];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) a[3 + i * 3 + j] += T[0][i] * ST[0][j] + T[1][i] * ST[1][j] + T[2][i] * ST[2][j];
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) a[j] += mb[j];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) a[3 + i * 3 + j] += S[i][j];
        }
    }
    // the twelve sums together: wave trees by shuffles, then the NT/64 wave partials through LDS in block_sum's order —
    // two barriers instead of twenty-four
#pragma unroll
    for (int i = 0; i < 12; ++i) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) a[i] += __shfl_xor(a[i], m);
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) red[(tid >> 6) * 12 + i] = a[i];
    }
    __syncthreads();
    if (tid < 12) {
        double r = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) r += red[w * 12 + tid];
        tot[tid] = r;
    }
    __syncthreads();
    double mx[3], Cx[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) mx[j] = tot[j] / M;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Cx[i][j] = tot[3 + i * 3 + j] / M - mx[i] * mx[j];
    if (tid < 3) stats[tid] = mx[tid];
    if (tid < 9) stats[3 + tid] = Cx[tid / 3][tid % 3];
    if (tid < 64) {
        const int c = tid;
        const double w[3] = {(double)w1[c * 3], (double)w1[c * 3 + 1], (double)w1[c * 3 + 2]};
        const double mu = w[0] * mx[0] + w[1] * mx[1] + w[2] * mx[2] + (double)b1[c];
        double var = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) var += w[i] * Cx[i][j] * w[j];
        var = var > 0.0 ? var : 0.0;
        const double is = 1.0 / sqrt(var + eps);
        const double sc = (double)g1[c] * is;
        chan[c] = (float)sc;
        chan[64 + c] = (float)((double)be1[c] - mu * sc);
        chan[128 + c] = (float)is;
        chan[192 + c] = (float)(-mu * is);
        stats[12 + c] = mu;
        stats[76 + c] = var;
        running_update(rm, rv, c, mu, var, M, momentum);
    }
    if (tid == 0 && nbt) *nbt += 1;
}

// ---------------------------------------------------------------------------------------
// BN2: tot (128,2) f64 = sum / sum of squares of z2 (pngpd_reduce_partials of the pass-B partials)
//   -> chan (4,128) f32 = s2c,t2c,is2,nm2 ; stats f64 = mu2r[128], var2[128]
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void bn2_fin(int c, double sum, double sq, double M, const float *__restrict__ b2,
                                        const float *__restrict__ g2, const float *__restrict__ be2, double eps,
                                        double momentum, float *rm, float *rv, long long *nbt,
                                        float *__restrict__ chan, double *__restrict__ stats) {
    const double mu = sum / M;
    double var = sq / M - mu * mu;
    var = var > 0.0 ? var : 0.0;
    const double is = 1.0 / sqrt(var + eps);
    const double sc = (double)g2[c] * is;
    chan[c] = (float)sc;
    chan[128 + c] = (float)((double)be2[c] - mu * sc);
    chan[256 + c] = (float)is;
    chan[384 + c] = (float)(-mu * is);
    stats[c] = mu;
    stats[128 + c] = var;
    running_update(rm, rv, c, mu + (double)b2[c], var, M, momentum);
    if (c == 0 && nbt) *nbt += 1;
}

__global__ __launch_bounds__(128) void bn2_finalize_kernel(
    const double *__restrict__ tot, double M, const float *__restrict__ b2,
    const float *__restrict__ g2, const float *__restrict__ be2, double eps, double momentum,
    float *rm, float *rv, long long *nbt, float *__restrict__ chan, double *__restrict__ stats) {
    const int c = threadIdx.x;
    bn2_fin(c, tot[c * 2], tot[c * 2 + 1], M, b2, g2, be2, eps, momentum, rm, rv, nbt, chan, stats);
}

// ---------------------------------------------------------------------------------------
// BN3: tot (2,1024) f64 (pngpd_reduce_partials of the pass-C psum) -> stats f64 = mu3s[1024], var3[1024]
// (of the sign-folded z3s)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void bn3_fin(int c, double sum, double sq, double M, const float *__restrict__ b3,
                                        const float *__restrict__ g3, double momentum, float *rm, float *rv,
                                        long long *nbt, double *__restrict__ stats) {
    const double mu = sum / M;
    double var = sq / M - mu * mu;
    var = var > 0.0 ? var : 0.0;
    stats[c] = mu;
    stats[1024 + c] = var;
    const double sgn = g3[c] >= 0.f ? 1.0 : -1.0;
    running_update(rm, rv, c, sgn * mu + (double)b3[c], var, M, momentum);
    if (c == 0 && nbt) *nbt += 1;
}

__global__ __launch_bounds__(NT) void bn3_finalize_kernel(
    const double *__restrict__ tot /* (2,1024) */, double M, const float *__restrict__ b3,
    const float *__restrict__ g3, double momentum, float *rm, float *rv, long long *nbt,
    double *__restrict__ stats) {
    const int c = blockIdx.x * NT + threadIdx.x;
    bn3_fin(c, tot[c], tot[1024 + c], M, b3, g3, momentum, rm, rv, nbt, stats);
}

// pooled[b][c] = (relu?)(g3 * zhat + be3), zhat = sgn*(max_s pmax - mu3s)/sig3 ; idx = arg of the max
// parg and idx carry no __restrict__: the pool refinement calls this in place (parg == idx, S == 1 — every thread reads
// its own element before it rewrites the same value), which a restrict contract would make undefined behaviour
__global__ __launch_bounds__(NT) void pool_finalize_kernel(
    const float *__restrict__ pmax, const int *parg, int S, const double *__restrict__ stats,
    const float *__restrict__ g3, const float *__restrict__ be3, double eps, int relu_last, int total,
    float *__restrict__ pooled, int *idx, float *__restrict__ zhat) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    const int b = i >> 10, c = i & 1023;
    const float *p = pmax + (size_t)b * S * 1024 + c;
    const int *a = parg + (size_t)b * S * 1024 + c;
    float m = p[0]; int am = a[0];
    for (int s = 1; s < S; ++s) {
        const float v = p[(size_t)s * 1024];
        if (v > m) { m = v; am = a[(size_t)s * 1024]; }   // strict: the earliest split wins ties
    }
    const double sgn = g3[c] >= 0.f ? 1.0 : -1.0;
    const double zh = sgn * ((double)m - stats[c]) / sqrt(stats[1024 + c] + eps);
    double y = (double)g3[c] * zh + (double)be3[c];
    if (relu_last && !(y > 0.0)) y = 0.0;
    pooled[i] = (float)y;
    idx[i] = am;
    zhat[i] = (float)zh;
}

// ---------------------------------------------------------------------------------------
// backward of BN3 affine + the sparse-term weights:
//   d = dp (masked by pooled>0 if relu_last);  dbe3 = sum_b d;  dg3 = sum_b d*zhat
//   coef[b][c] = d * g3/sig3 ;  m12 f64 = m1[1024] (= dbe3/M), m2[1024] (= dg3/M)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(16 * PNGPD_BN3_RL) void bn3_bwd_prep_kernel(
    const float *__restrict__ dp, const float *__restrict__ pooled, const float *__restrict__ zhat, int B,
    double M, const float *__restrict__ g3, const double *__restrict__ stats, double eps, int relu_last,
    float *__restrict__ coef, float *__restrict__ dg3, float *__restrict__ dbe3, double *__restrict__ m12) {
    // block = 16 channels x 64 row lanes (64 workgroups); the row loop is unrolled so that 8 rows' loads are in flight
    __shared__ double r1[PNGPD_BN3_RL][17], r2[PNGPD_BN3_RL][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    const double s3 = (double)g3[c] / sqrt(stats[1024 + c] + eps);
    double a1 = 0.0, a2 = 0.0;
#pragma unroll 8
    for (int b = ry; b < B; b += PNGPD_BN3_RL) {
        const size_t i = (size_t)b * 1024 + c;
        float d = dp[i];
        if (relu_last && !(pooled[i] > 0.f)) d = 0.f;
        a1 += (double)d;
        a2 += (double)d * (double)zhat[i];
        coef[i] = (float)((double)d * s3);
    }
    r1[ry][cx] = a1; r2[ry][cx] = a2;
    __syncthreads();
    if (ry == 0) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll 8
        for (int i = 0; i < PNGPD_BN3_RL; ++i) { t1 += r1[i][cx]; t2 += r2[i][cx]; }
        dbe3[c] = (float)t1;
        dg3[c] = (float)t2;
        m12[c] = t1 / M;
        m12[1024 + c] = t2 / M;
    }
}

// out[o][j] = sum_r in[o][r][j]   (fp32 partials -> fp64), deterministic order.
// block = 32 columns x 32 row-lanes (rows strided by 32, then a fixed-order LDS tree over the lanes).
__global__ __launch_bounds__(32 * PNGPD_RED_RL) void reduce_partials_kernel(const float *__restrict__ in, int R, int n,
                                                              double *__restrict__ out) {
    __shared__ double red[PNGPD_RED_RL][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + cx;
    double s = 0.0;
    if (j < n) {
        const float *p = in + (size_t)blockIdx.y * R * n + j;
        #pragma unroll 8   // eight rows' loads in flight per thread: the kernel is a latency chain otherwise (2-3 TB/s)
        for (int r = ry; r < R; r += PNGPD_RED_RL) s += (double)p[(size_t)r * n];
    }
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && j < n) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < PNGPD_RED_RL; ++i) t += red[i][cx];
        out[(size_t)blockIdx.y * n + j] = t;
    }
}

// Up to four such reductions in ONE launch (the partial buffers of one training pass are reduced together:
// 4 launches per trunk instead of 9).  Segment g covers blocks [first[g], first[g+1]).
struct ReduceSeg { const float *in; double *out; int outer, R, n, blocks_per_outer; };
struct ReduceSegs { ReduceSeg seg[4]; int first[5]; };

__global__ __launch_bounds__(32 * PNGPD_RED_RL) void reduce_partials_multi_kernel(ReduceSegs A) {
    __shared__ double red[PNGPD_RED_RL][33];
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) g += ((int)blockIdx.x >= A.first[i]) ? 1 : 0;
    const ReduceSeg sg = A.seg[g];
    const int lb = blockIdx.x - A.first[g];
    const int o = lb / sg.blocks_per_outer, jb = lb - o * sg.blocks_per_outer;
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int j = jb * 32 + cx;
    double s = 0.0;
    if (j < sg.n) {
        const float *p = sg.in + (size_t)o * sg.R * sg.n + j;
        #pragma unroll 8   // eight rows' loads in flight per thread: the kernel is a latency chain otherwise (2-3 TB/s)
        for (int r = ry; r < sg.R; r += PNGPD_RED_RL) s += (double)p[(size_t)r * sg.n];
    }
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && j < sg.n) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < PNGPD_RED_RL; ++i) t += red[i][cx];
        sg.out[(size_t)o * sg.n + j] = t;
    }
}

// Element (k, j) of S2 = sum_points h2 h2^T from the 10 accumulator blocks pass D keeps (S2c f64 [12][16][64]:
// slot 3w+q of wave w = block (w, (w+q) mod 4) for q < 2; slot 3w+2 = the partial of block (w, w+2) [w < 2] or of
// block (w-2, w) [w >= 2] over half of each tile's points — the two are added here; raw MFMA register layout
// D[i = (r&3) + 8 (r>>2) + 4 (lane>>5)][j = lane & 31]); the other six blocks are transposes.
__device__ __forceinline__ double s2_at(const double *__restrict__ S2c, int k, int j) {
    int a = k >> 5, b = j >> 5, i = k & 31, jj = j & 31;
    const int d = (b - a) & 3;
    int q = d;
    if (d == 3 || (d == 2 && a >= 2)) {   // stored as the transposed block (b, a)
        const int t = a; a = b; b = t;
        const int u = i; i = jj; jj = u;
        q = (b - a) & 3;
    }
    const int r = (i & 3) + 4 * (i >> 3), h = (i >> 2) & 1;
    const size_t e = (size_t)r * 64 + h * 32 + jj;
    double v = S2c[(size_t)(a * 3 + q) * 1024 + e];
    if (q == 2) v += S2c[(size_t)((a + 2) * 3 + 2) * 1024 + e];   // block (a, a+2): wave a+2 holds the other half of the points
    return v;
}

// dw3_finalize_kernel / a_cvec_finalize_kernel: bodies in pngpd_glue_bodies.h (pass E and the gather pass carry the
// same bodies as tail workgroups in the fused backward)
__global__ __launch_bounds__(128 * DW3_KQ) void dw3_finalize_kernel(const DW3Args A) {
    __shared__ double lds[DW3_LDS_DOUBLES];
    dw3_finalize_body<DW3_KQ>(A, (int)blockIdx.x, lds);
}

__global__ __launch_bounds__(512) void a_cvec_finalize_kernel(const ACvecArgs A) {
    __shared__ double lds[ACVEC_LDS_DOUBLES];
    a_cvec_finalize_body<512>(A, (int)blockIdx.x, lds);
}

// a12 (128,2) f64 = sum g2, sum g2*zhat2  ->  dg2 = a2, dbe2 = a1 and the pass-E vectors
//   evec (3,128) f32 = a1/M, a2/M, g2/sig2.
__device__ __forceinline__ void e_prep_fin(int o, double a1, double a2, double M, const float *__restrict__ g2,
                                           const double *__restrict__ stats2, double eps, float *__restrict__ dg2,
                                           float *__restrict__ dbe2, float *__restrict__ evec) {
    dg2[o] = (float)a2; dbe2[o] = (float)a1;
    evec[o] = (float)(a1 / M); evec[128 + o] = (float)(a2 / M);
    evec[256 + o] = (float)((double)g2[o] / sqrt(stats2[128 + o] + eps));
}

__global__ __launch_bounds__(128) void bwd_e_prep_kernel(
    const double *__restrict__ a12, double M, const float *__restrict__ g2, const double *__restrict__ stats2,
    double eps, float *__restrict__ dg2, float *__restrict__ dbe2, float *__restrict__ evec) {
    const int o = threadIdx.x;
    e_prep_fin(o, a12[o * 2], a12[o * 2 + 1], M, g2, stats2, eps, dg2, dbe2, evec);
}

// dW1[c][j] = s1 (Rp[c][j] - c1 mx[j] - (c2/sig1) (W1 Cx)[c][j]),  Rp = sum_b Rb[b] T_b  (or sum_b Rb[b])
// block = channel c, NT threads over b.   Rb f64 (B,64,3).
__device__ __forceinline__ void dw1_finalize_body(
    int c, const double *__restrict__ Rb, const float *__restrict__ trans, int B, const double *__restrict__ c12 /*(64,2)*/,
    const double *__restrict__ stats1, const float *__restrict__ w1, const float *__restrict__ g1, double eps,
    float *__restrict__ dW1, float *__restrict__ dg1, float *__restrict__ dbe1, double *red /* [NT] */) {
    const int tid = threadIdx.x;
    double r[3] = {0.0, 0.0, 0.0};
    for (int b = tid; b < B; b += NT) {
        const double *rb = Rb + ((size_t)b * 64 + c) * 3;
        if (trans) {
            const float *T = trans + (size_t)b * 9;
#pragma unroll
            for (int j = 0; j < 3; ++j) r[j] += rb[0] * (double)T[j] + rb[1] * (double)T[3 + j] + rb[2] * (double)T[6 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) r[j] += rb[j];
        }
    }
    double Rp[3];
    for (int j = 0; j < 3; ++j) Rp[j] = block_sum(r[j], red);
    if (tid == 0) {
        const double *mx = stats1, *Cx = stats1 + 3;
        const double sig = sqrt(stats1[76 + c] + eps);
        const double s1 = (double)g1[c] / sig;
        const double c1 = c12[c * 2], c2 = c12[c * 2 + 1];
        const double w[3] = {(double)w1[c * 3], (double)w1[c * 3 + 1], (double)w1[c * 3 + 2]};
        for (int j = 0; j < 3; ++j) {
            const double wcx = w[0] * Cx[0 * 3 + j] + w[1] * Cx[1 * 3 + j] + w[2] * Cx[2 * 3 + j];
            dW1[c * 3 + j] = (float)(s1 * (Rp[j] - c1 * mx[j] - (c2 / sig) * wcx));
        }
        dg1[c] = (float)c2; dbe1[c] = (float)c1;
    }
}

// dT_b = Y_b W1,  Y_b[i][c] = s1 (Rb[b][c][i] - m_b[i] c1/M - term3 c2/M),
//   term3 = ((S_b T_b)[i][:] . W1[c] + m_b[i] (b1 - mu1)) / sig1.     64 threads = c per cloud, NT/64 clouds per workgroup
__device__ __forceinline__ void dtrans_finalize_body(
    int blk, int B, const double *__restrict__ Rb, const float *__restrict__ trans, const double *__restrict__ mom,
    double M, const double *__restrict__ c12, const double *__restrict__ stats1, const float *__restrict__ w1,
    const float *__restrict__ b1, const float *__restrict__ g1, double eps, float *__restrict__ dT,
    double *lds /* [NT/64][9][64] */) {
    const int cl = threadIdx.x >> 6, c = threadIdx.x & 63;
    const int b = blk * (NT / 64) + cl;
    const bool live = b < B;
    double (*red)[64] = (double (*)[64])(lds + (size_t)cl * 9 * 64);
    if (live) {
        const double *m = mom + (size_t)b * 9;
        const double mb[3] = {m[0], m[1], m[2]};
        const double S[3][3] = {{m[3], m[4], m[5]}, {m[4], m[6], m[7]}, {m[5], m[7], m[8]}};
        double T[3][3];
#pragma unroll
        for (int i = 0; i < 9; ++i) T[i / 3][i % 3] = (double)trans[(size_t)b * 9 + i];
        const double w[3] = {(double)w1[c * 3], (double)w1[c * 3 + 1], (double)w1[c * 3 + 2]};
        const double sig = sqrt(stats1[76 + c] + eps);
        const double s1 = (double)g1[c] / sig;
        const double c1 = c12[c * 2] / M, c2 = c12[c * 2 + 1] / M;
        const double bmu = (double)b1[c] - stats1[12 + c];
        const double *rb = Rb + ((size_t)b * 64 + c) * 3;
        double Y[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double st = 0.0;   // (S T)[i][:] . w
#pragma unroll
            for (int j = 0; j < 3; ++j) st += (S[i][0] * T[0][j] + S[i][1] * T[1][j] + S[i][2] * T[2][j]) * w[j];
            const double term3 = (st + mb[i] * bmu) / sig;
            Y[i] = s1 * (rb[i] - mb[i] * c1 - term3 * c2);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) red[i * 3 + j][c] = Y[i] * w[j];
    }
    __syncthreads();
    if (live && c < 9) {
        double s = 0.0;
        for (int k = 0; k < 64; ++k) s += red[c][k];
        dT[(size_t)b * 9 + c] = (float)s;
    }
}

// Both finalizes of pass E's sums in ONE launch (they are independent): blocks [0, 64) = dW1 / dg1 / dbe1 per channel,
// the rest (dT != NULL) = dT for NT/64 clouds each.
__global__ __launch_bounds__(NT) void dw1_dtrans_finalize_kernel(
    const double *__restrict__ Rb, const float *__restrict__ trans, const double *__restrict__ mom, int B, double M,
    const double *__restrict__ c12, const double *__restrict__ stats1, const float *__restrict__ w1,
    const float *__restrict__ b1, const float *__restrict__ g1, double eps, float *__restrict__ dW1,
    float *__restrict__ dg1, float *__restrict__ dbe1, float *__restrict__ dT) {
    __shared__ double lds[(NT / 64) * 9 * 64];
    if (blockIdx.x < 64)
        dw1_finalize_body((int)blockIdx.x, Rb, trans, B, c12, stats1, w1, g1, eps, dW1, dg1, dbe1, lds);
    else
        dtrans_finalize_body((int)blockIdx.x - 64, B, Rb, trans, mom, M, c12, stats1, w1, b1, g1, eps, dT, lds);
}

// ---------------------------------------------------------------------------------------
// Fused reduce + finalize: the deterministic fp64 reduction of up to four partial buffers of ONE training pass
// (the reduce_partials_multi_kernel scheme: 32 columns x 32 row lanes per workgroup, fixed-order LDS tree) with the
// per-channel finalize of the pass applied by the workgroup that owns the channel's totals — one launch where the
// step used to make two to four (reduce, finalize, cast, fills).  The arithmetic of every output is the arithmetic
// of the separate kernels above (same reduction order, same bn2_fin / bn3_fin / e_prep_fin), so both ways of
// sequencing a step give bit-identical results.
//   RF_F64 / RF_F32  out[o][j] = sum_r in[o][r][j]            (fp64 / rounded once to fp32)
//   RF_BN2           in (blk,128,2) pass-B partials -> bn2_fin per channel      (pngpd_bn2_finalize)
//   RF_BN3           in (blk,2,1024) pass-C partials -> bn3_fin per channel     (pngpd_bn3_finalize)
//   RF_EPREP         in (blk,128,2) pass-D partials -> e_prep_fin per channel   (pngpd_bwd_e_prep)
//   RF_ZERO          no input: f0[64] = f1[128] = f2[1024] = 0 (the exactly-zero gradients of conv biases ahead of a
//                    train-mode BatchNorm)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(32 * PNGPD_RED_RL) void reduce_fin_kernel(RFArgs A) {
    __shared__ double red[PNGPD_RED_RL][33];
    __shared__ double tot[32];
    int g = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) g += ((int)blockIdx.x >= A.first[i]) ? 1 : 0;
    const RFSeg sg = A.seg[g];
    const int lb = blockIdx.x - A.first[g];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    if (sg.kind == RF_ZERO) {
        const int t = threadIdx.x;
        if (t < 64 && sg.f0) sg.f0[t] = 0.f;
        if (t < 128 && sg.f1) sg.f1[t] = 0.f;
        if (sg.f2) sg.f2[t] = 0.f;
        return;
    }
    const int o = lb / sg.bpo, jb = lb - o * sg.bpo;
    if (sg.vec) {
        // wide variant for the big plain reductions (Gram blocks, dW2 shares, the gathered G): a workgroup owns 128
        // consecutive columns and every thread FOUR of them (one 16-byte load per row, 512 contiguous bytes per row and
        // workgroup instead of 128).  Per column the arithmetic is unchanged — row lane ry adds rows ry, ry + RL, ... in
        // order, then the lanes are added in order — so the result is bit-identical to the narrow path.
        const int j4 = jb * 128 + cx * 4;
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        const f32x4 *p = (const f32x4 *)(sg.in + (size_t)o * sg.R * sg.n + j4);
        const size_t rs = (size_t)sg.n >> 2;
        #pragma unroll 8
        for (int r = ry; r < sg.R; r += PNGPD_RED_RL) {
            const f32x4 v = p[(size_t)r * rs];
            s4[0] += (double)v[0]; s4[1] += (double)v[1]; s4[2] += (double)v[2]; s4[3] += (double)v[3];
        }
        double t4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e) __syncthreads();
            red[ry][cx] = s4[e];
            __syncthreads();
            if (ry == 0) {
                double a = 0.0;
#pragma unroll
                for (int i = 0; i < PNGPD_RED_RL; ++i) a += red[i][cx];
                t4[e] = a;
            }
        }
        if (ry == 0) {
            if (sg.kind == RF_F64) {
                double *out = (double *)sg.out + (size_t)o * sg.n + j4;
#pragma unroll
                for (int e = 0; e < 4; ++e) out[e] = t4[e];
            } else {
                *(f32x4 *)((float *)sg.out + (size_t)o * sg.n + j4) = f32x4{(float)t4[0], (float)t4[1], (float)t4[2], (float)t4[3]};
            }
        }
        return;
    }
    const int j = jb * 32 + cx;
    const int planes = sg.kind == RF_BN3 ? 2 : 1;
    const size_t stride = (size_t)sg.n * planes;
    double t[2] = {0.0, 0.0};
    for (int pl = 0; pl < planes; ++pl) {
        double s = 0.0;
        if (j < sg.n) {
            const float *p = sg.in + (size_t)o * sg.R * stride + (size_t)pl * sg.n + j;
            #pragma unroll 8   // eight rows' loads in flight per thread: the kernel is a latency chain otherwise (2-3 TB/s)
            for (int r = ry; r < sg.R; r += PNGPD_RED_RL) s += (double)p[(size_t)r * stride];
        }
        if (pl) __syncthreads();
        red[ry][cx] = s;
        __syncthreads();
        if (ry == 0) {
            double a = 0.0;
#pragma unroll
            for (int i = 0; i < PNGPD_RED_RL; ++i) a += red[i][cx];
            t[pl] = a;
        }
    }
    if (sg.kind == RF_F64) {
        if (ry == 0 && j < sg.n) ((double *)sg.out)[(size_t)o * sg.n + j] = t[0];
    } else if (sg.kind == RF_F32) {
        if (ry == 0 && j < sg.n) ((float *)sg.out)[(size_t)o * sg.n + j] = (float)t[0];
    } else if (sg.kind == RF_BN3) {
        if (ry == 0 && j < sg.n) bn3_fin(j, t[0], t[1], A.M, sg.p0, sg.p1, A.momentum, sg.rm, sg.rv, sg.nbt, sg.s0);
    } else {   // RF_BN2 / RF_EPREP: columns (2c, 2c+1) = the two sums of channel c
        if (ry == 0) tot[cx] = t[0];
        __syncthreads();
        if (ry == 0 && cx < 16) {
            const int c = jb * 16 + cx;
            if (sg.kind == RF_BN2)
                bn2_fin(c, tot[2 * cx], tot[2 * cx + 1], A.M, sg.p0, sg.p1, sg.p2, A.eps, A.momentum, sg.rm, sg.rv,
                        sg.nbt, sg.f0, sg.s0);
            else
                e_prep_fin(c, tot[2 * cx], tot[2 * cx + 1], A.M, sg.p0, sg.d0, A.eps, sg.f0, sg.f1, sg.f2);
        }
    }
}

int pngpd_reduce_fin_launch(RFArgs &A, int nseg, void *stream) {
    int total = 0;
    for (int g = 0; g < 4; ++g) {
        A.first[g] = total;
        if (g >= nseg) { A.seg[g] = RFSeg{}; A.seg[g].bpo = 1; continue; }
        RFSeg &s = A.seg[g];
        if (s.kind == RF_ZERO) { s.bpo = 1; total += 1; continue; }
        if (!s.in || s.outer <= 0 || s.R <= 0 || s.n <= 0) return PNGPD_ERR_INVALID_ARG;
        s.vec = ((s.kind == RF_F64 || s.kind == RF_F32) && (s.n & 127) == 0 && s.n >= 1024 &&
                 (((uintptr_t)s.in | (uintptr_t)s.out) & 15) == 0) ? 1 : 0;
        s.bpo = s.vec ? s.n / 128 : (s.n + 31) / 32;
        total += s.outer * s.bpo;
    }
    A.first[4] = total;
    if (total == 0) return PNGPD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(reduce_fin_kernel, dim3(total), dim3(32 * PNGPD_RED_RL), 0, (hipStream_t)stream, A);
    return pngpd_launch_status();
}

// ---------------------------------------------------------------------------------------
// Weight re-layout for the training passes, every matrix of a trunk in ONE launch (the step used to make one
// fold_conv_bn / split_pack launch per matrix plus a transposed copy and a sign tensor):
//   logical matrix Mx (C,K) = W (transpose: W is (K,C) row-major and Mx = W^T; src_packed: W is the MFMA_B-packed
//   128x128 matrix pngpd_a_cvec_finalize writes), rows scaled by sign(sgn_src[c]) (+1 for >= 0: the sign fold of W3),
//   written MFMA_B-packed fp32 (fmt 0, pngpd_fold_conv_bn's layout) or as split_pack_bf16 fragments (fmt 1).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void train_pack_kernel(PackArgs A) {
    if ((int)blockIdx.x >= A.first[PACK_MAX_JOBS]) {     // pass A rides behind the pack jobs (workgroup-uniform branch)
        cloud_moments_body(A.mom_x, A.mom_N, (int)blockIdx.x - A.first[PACK_MAX_JOBS], A.mom);
        return;
    }
    int g = 0;
#pragma unroll
    for (int i = 1; i < PACK_MAX_JOBS; ++i) g += ((int)blockIdx.x >= A.first[i]) ? 1 : 0;
    const PackJob jb = A.job[g];
    const int idx = (blockIdx.x - A.first[g]) * 256 + threadIdx.x;
    if (idx >= jb.C * jb.K) return;
    const int c = idx / jb.K, k = idx - c * jb.K;
    float v;
    if (jb.src_packed) {
        const int cb = c >> 5, jj = c & 31, kb = k >> 3, h = (k >> 2) & 1, t = k & 3;
        v = jb.W[(((cb * (jb.K >> 3) + kb) * 64) + h * 32 + jj) * 4 + t];
    } else {
        v = jb.transpose ? jb.W[(size_t)k * jb.C + c] : jb.W[idx];
    }
    if (jb.sgn_src) v = jb.sgn_src[c] >= 0.f ? v : -v;
    if (jb.fmt == 0) {
        const int cb = c >> 5, j = c & 31, kb = k >> 3, h = (k >> 2) & 1, t = k & 3;
        ((float *)jb.out)[(((cb * (jb.K >> 3) + kb) * 64) + h * 32 + j) * 4 + t] = v;
    } else {
        const unsigned short hi = __builtin_bit_cast(unsigned short, (__bf16)v);
        const float hv = __uint_as_float(((unsigned)hi) << 16);
        const unsigned short lo = __builtin_bit_cast(unsigned short, (__bf16)(v - hv));
        const int cb = c >> 5, j = c & 31, ks = k >> 4, h = (k >> 3) & 1, t = k & 7, KS = jb.K >> 4;
        const size_t base = ((size_t)(cb * KS + ks) * 2) * 64 * 8 + (size_t)(h * 32 + j) * 8 + t;
        ((unsigned short *)jb.out)[base] = hi;
        ((unsigned short *)jb.out)[base + 64 * 8] = lo;
    }
}

int pngpd_train_pack_launch(PackArgs &A, int njobs, void *stream) {
    if (njobs <= 0 || njobs > PACK_MAX_JOBS) return PNGPD_ERR_INVALID_ARG;
    int total = 0;
    for (int g = 0; g < PACK_MAX_JOBS; ++g) {
        A.first[g] = total;
        if (g >= njobs) { A.job[g] = PackJob{}; continue; }
        const PackJob &j = A.job[g];
        if (!j.W || !j.out || j.C <= 0 || j.K <= 0 || (j.C & 31) || (j.K & (j.fmt ? 15 : 7))) return PNGPD_ERR_INVALID_ARG;
        total += (j.C * j.K + 255) / 256;
    }
    A.first[PACK_MAX_JOBS] = total;
    if (A.mom_x) {
        if (!A.mom || A.mom_N <= 0 || A.mom_B <= 0) return PNGPD_ERR_INVALID_ARG;
        total += A.mom_B;
    }
    hipLaunchKernelGGL(train_pack_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, A);
    return pngpd_launch_status();
}

// ---------------------------------------------------------------------------------------
// Adam over ONE flat parameter / gradient / moment buffer (main_1v.py:61 `optim.Adam(model.parameters(), lr)` —
// torch's defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad), the single-tensor update of
// torch/optim/adam.py in fp32:
//   m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// lr_dev / step_dev (nullable) are device-resident values for captured graphs (StepLR rewrites lr in place;
// step_dev holds the step count t of THIS update as a float, advanced by pngpd_adam_step_inc).  The gradient is first
// multiplied by gscale and, if gdiv_dev is given, divided by max(*gdiv_dev, 1) — the all-reduced kept-sample count of
// a data-parallel step, which never has to visit the host.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_flat_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v, long n,
                                                        float lr_host, const float *__restrict__ lr_dev, float b1,
                                                        float b2, float eps, float step_host,
                                                        const float *__restrict__ step_dev, float gscale,
                                                        const float *__restrict__ gdiv_dev) {
    if (gdiv_dev) gscale /= fmaxf(*gdiv_dev, 1.f);
    const float lr = lr_dev ? *lr_dev : lr_host;
    const float t = step_dev ? *step_dev : step_host;
    const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
    const float step_size = lr / bc1;
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 + 4 <= n) {
        f32x4 pv = *(f32x4 *)(p + i0), gv = *(const f32x4 *)(g + i0), mv = *(f32x4 *)(m + i0), vv = *(f32x4 *)(v + i0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = gv[e] * gscale;
            mv[e] = mv[e] + (gg - mv[e]) * (1.f - b1);
            vv[e] = vv[e] * b2 + (1.f - b2) * gg * gg;
            pv[e] = pv[e] - step_size * (mv[e] / (sqrtf(vv[e]) / bc2s + eps));
        }
        *(f32x4 *)(p + i0) = pv; *(f32x4 *)(m + i0) = mv; *(f32x4 *)(v + i0) = vv;
    } else {
        for (long i = i0; i < n; ++i) {
            const float gg = g[i] * gscale;
            const float mm = m[i] + (gg - m[i]) * (1.f - b1);
            const float vq = v[i] * b2 + (1.f - b2) * gg * gg;
            m[i] = mm; v[i] = vq;
            p[i] = p[i] - step_size * (mm / (sqrtf(vq) / bc2s + eps));
        }
    }
}

__global__ void adam_step_inc_kernel(float *step) { *step += 1.f; }

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
#define LAUNCH(kernel, grid, block, ...) \
    hipLaunchKernelGGL(kernel, grid, block, 0, (hipStream_t)stream, __VA_ARGS__); \
    return pngpd_launch_status()

extern "C" {

int pngpd_bn1_finalize(const double *mom, const float *trans, int B, int N, const float *w1, const float *b1,
                       const float *g1, const float *be1, float eps, float momentum, float *rm, float *rv,
                       long long *nbt, float *chan, double *stats, void *stream) {
    if (!mom || !w1 || !b1 || !g1 || !be1 || !chan || !stats || B <= 0 || N <= 0) return PNGPD_ERR_INVALID_ARG;
    LAUNCH(bn1_finalize_kernel, dim3(1), dim3(NT), mom, trans, B, (double)B * N, w1, b1, g1, be1, (double)eps,
           (double)momentum, rm, rv, nbt, chan, stats);
}

int pngpd_bn2_finalize(const double *tot, int B, int N, const float *b2, const float *g2,
                       const float *be2, float eps, float momentum, float *rm, float *rv, long long *nbt,
                       float *chan, double *stats, void *stream) {
    if (!tot || !b2 || !g2 || !be2 || !chan || !stats || B <= 0 || N <= 0) return PNGPD_ERR_INVALID_ARG;
    LAUNCH(bn2_finalize_kernel, dim3(1), dim3(128), tot, (double)B * N, b2, g2, be2, (double)eps,
           (double)momentum, rm, rv, nbt, chan, stats);
}

int pngpd_bn3_finalize(const double *tot, int B, int N, const float *b3, const float *g3,
                       float momentum, float *rm, float *rv, long long *nbt, double *stats, void *stream) {
    if (!tot || !b3 || !g3 || !stats || B <= 0 || N <= 0) return PNGPD_ERR_INVALID_ARG;
    LAUNCH(bn3_finalize_kernel, dim3(1024 / NT), dim3(NT), tot, (double)B * N, b3, g3, (double)momentum,
           rm, rv, nbt, stats);
}

int pngpd_pool_finalize(const float *pmax, const int *parg, int B, int S, const double *stats, const float *g3,
                        const float *be3, float eps, int relu_last, float *pooled, int *idx, float *zhat,
                        void *stream) {
    if (!pmax || !parg || !stats || !g3 || !be3 || !pooled || !idx || !zhat || B <= 0 || S <= 0)
        return PNGPD_ERR_INVALID_ARG;
    const int total = B * 1024;
    LAUNCH(pool_finalize_kernel, dim3((total + NT - 1) / NT), dim3(NT), pmax, parg, S, stats, g3, be3, (double)eps,
           relu_last, total, pooled, idx, zhat);
}

int pngpd_bn3_bwd_prep(const float *dp, const float *pooled, const float *zhat, int B, int N, const float *g3,
                       const double *stats, float eps, int relu_last, float *coef, float *dg3, float *dbe3,
                       double *m12, void *stream) {
    if (!dp || !pooled || !zhat || !g3 || !stats || !coef || !dg3 || !dbe3 || !m12 || B <= 0 || N <= 0)
        return PNGPD_ERR_INVALID_ARG;
    LAUNCH(bn3_bwd_prep_kernel, dim3(64), dim3(16 * PNGPD_BN3_RL), dp, pooled, zhat, B, (double)B * N, g3, stats,
           (double)eps, relu_last, coef, dg3, dbe3, m12);
}

int pngpd_reduce_partials(const float *in, int outer, int R, int n, double *out, void *stream) {
    if (!in || !out || outer <= 0 || R <= 0 || n <= 0) return PNGPD_ERR_INVALID_ARG;
    LAUNCH(reduce_partials_kernel, dim3((n + 31) / 32, outer), dim3(32 * PNGPD_RED_RL), in, R, n, out);
}

int pngpd_reduce_partials4(const float *in0, int outer0, int R0, int n0, double *out0,
                           const float *in1, int outer1, int R1, int n1, double *out1,
                           const float *in2, int outer2, int R2, int n2, double *out2,
                           const float *in3, int outer3, int R3, int n3, double *out3, void *stream) {
    const float *in[4] = {in0, in1, in2, in3};
    double *out[4] = {out0, out1, out2, out3};
    const int outer[4] = {outer0, outer1, outer2, outer3}, R[4] = {R0, R1, R2, R3}, n[4] = {n0, n1, n2, n3};
    ReduceSegs A;
    int total = 0;
    for (int g = 0; g < 4; ++g) {
        A.first[g] = total;
        A.seg[g].in = in[g]; A.seg[g].out = out[g];
        A.seg[g].outer = 0; A.seg[g].R = 0; A.seg[g].n = 0; A.seg[g].blocks_per_outer = 1;
        if (!in[g]) continue;   // unused slot
        if (!out[g] || outer[g] <= 0 || R[g] <= 0 || n[g] <= 0) return PNGPD_ERR_INVALID_ARG;
        A.seg[g].outer = outer[g]; A.seg[g].R = R[g]; A.seg[g].n = n[g];
        A.seg[g].blocks_per_outer = (n[g] + 31) / 32;
        total += outer[g] * A.seg[g].blocks_per_outer;
    }
    A.first[4] = total;
    if (total == 0) return PNGPD_ERR_INVALID_ARG;
    LAUNCH(reduce_partials_multi_kernel, dim3(total), dim3(32 * PNGPD_RED_RL), A);
}

int pngpd_a_cvec_finalize(const double *sh, int B, int N, const float *w3, const float *g3, const double *stats,
                          const double *m12, float eps, float *Ap, float *cvec, void *stream) {
    if (!sh || !w3 || !g3 || !stats || !m12 || !Ap || !cvec || B <= 0 || N <= 0) return PNGPD_ERR_INVALID_ARG;
    const ACvecArgs A{w3, g3, stats, m12, sh, (double)B * N, (double)eps, Ap, cvec};
    LAUNCH(a_cvec_finalize_kernel, dim3(128), dim3(512), A);
}

int pngpd_dw3_finalize(const double *G, const double *S2c, const double *sh, int B, int N, const float *w3,
                       const float *g3, const double *stats, const double *m12, float eps, float *dW3,
                       void *stream) {
    if (!G || !S2c || !sh || !w3 || !g3 || !stats || !m12 || !dW3 || B <= 0 || N <= 0)
        return PNGPD_ERR_INVALID_ARG;
    const DW3Args A{G, S2c, sh, (double)B * N, w3, g3, stats, m12, (double)eps, dW3};
    LAUNCH(dw3_finalize_kernel, dim3(1024 / DW3_CPB), dim3(128 * DW3_KQ), A);
}

int pngpd_bwd_e_prep(const double *a12, int B, int N, const float *g2, const double *stats2, float eps,
                     float *dg2, float *dbe2, float *evec, void *stream) {
    if (!a12 || !g2 || !stats2 || !dg2 || !dbe2 || !evec || B <= 0 || N <= 0) return PNGPD_ERR_INVALID_ARG;
    LAUNCH(bwd_e_prep_kernel, dim3(1), dim3(128), a12, (double)B * N, g2, stats2, (double)eps, dg2, dbe2, evec);
}

int pngpd_dw1_finalize(const double *Rb, const float *trans, const double *mom, int B, int N, const double *c12,
                       const double *stats1, const float *w1, const float *b1, const float *g1, float eps,
                       float *dW1, float *dg1, float *dbe1, float *dT, void *stream) {
    if (!Rb || !mom || !c12 || !stats1 || !w1 || !b1 || !g1 || !dW1 || !dg1 || !dbe1 || B <= 0 || N <= 0)
        return PNGPD_ERR_INVALID_ARG;
    if (dT && !trans) return PNGPD_ERR_INVALID_ARG;
    const int nblk = 64 + (dT ? (B + NT / 64 - 1) / (NT / 64) : 0);
    LAUNCH(dw1_dtrans_finalize_kernel, dim3(nblk), dim3(NT), Rb, trans, mom, B, (double)B * N, c12, stats1, w1, b1, g1,
           (double)eps, dW1, dg1, dbe1, dT);
}

int pngpd_adam_flat(float *p, const float *g, float *m, float *v, long long n, float lr, const float *lr_dev,
                    float beta1, float beta2, float eps, float step, const float *step_dev, float gscale,
                    const float *gdiv_dev, void *stream) {
    if (!p || !g || !m || !v || n <= 0 || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15))
        return PNGPD_ERR_INVALID_ARG;
    if (!step_dev && step < 1.f) return PNGPD_ERR_INVALID_ARG;
    const long quads = ((long)n + 3) / 4;
    LAUNCH(adam_flat_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), p, g, m, v, (long)n, lr, lr_dev, beta1,
           beta2, eps, step, step_dev, gscale, gdiv_dev);
}

int pngpd_adam_step_inc(float *step_dev, void *stream) {
    if (!step_dev) return PNGPD_ERR_INVALID_ARG;
    LAUNCH(adam_step_inc_kernel, dim3(1), dim3(1), step_dev);
}

}  // extern "C"
