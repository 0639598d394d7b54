// libpngpd — the FC stacks at SMALL batches (B <= 128: the reference's own recipe, main_1v.py:160 `batch-size 64`, and
// the per-GPU share of a strong-scaled step), pointnet.py:35-43 / :191-194 under main_1v.py:72-76.
//
// At these sizes a Linear layer and the BatchNorm1d next to it are each one 5-10 us latency chain on a few workgroups,
// and the BatchNorm needs a column of ALL rows.  Here a workgroup owns all rows of a 32-channel column: VW = ceil(B/32)
// "virtual workgroups" of four waves each compute the column's 32x32 tiles exactly as the per-op kernels do (same K
// quarters per wave, same partial-tile sums: fc_tile / fc_bwd_tile of pngpd_fc_tile.h), then the first 256 threads run
// the column's BatchNorm in the per-op kernel's summation order (bn1d_*_tail).  No other workgroup is involved, so
// nothing is exchanged between workgroups — one launch does what two did, bit for bit:
//   forward    fc + BatchNorm(batch statistics) + ReLU             (pngpd_fc_fwd + pngpd_bn1d_fwd_train)
//   backward   fc_bwd of the NEXT layer + this BatchNorm's backward (pngpd_fc_bwd + pngpd_bn1d_bwd): the column
//              workgroups produce dx = dy of the BatchNorm for all rows and turn it into dz; the dW tiles of the
//              same launch ride in further workgroups, VW tiles each
// Five + six launches per stack and step become three + four (tools/ab_lib.sh: profiles/r04_*).
#include "pngpd_fc_tile.h"
#include "pngpd_internal.h"

namespace {

constexpr int COL_NR = 128 / BN1D_RL;   // rows per logical BatchNorm lane that cover B <= 128 (2; 8 in the sanitizer build)

template <int VW>
__global__ __launch_bounds__(VW * 256) void fc_bn_col_kernel(
    const float *__restrict__ in, int B, int K, const float *__restrict__ W, const float *__restrict__ bias, int Nout,
    const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float momentum, float *z, float *y,
    float *__restrict__ mean, float *__restrict__ var, float *rm, float *rv, long long *nbt) {
    __shared__ float lds[VW * FC_RED_FLOATS];
    static_assert(FC_RED_FLOATS >= BN1D_RL * 33, "the BatchNorm lanes reuse the first tile's partials");
    const int vw = threadIdx.x >> 8, cb = blockIdx.x;
    fc_tile<true>(in, B, K, W, bias, Nout, PNGPD_EPI_NONE, z, vw, cb, lds + vw * FC_RED_FLOATS, vw * 32 < B);
    __syncthreads();   // the column's z tiles are stored (and visible to this workgroup)
    bn1d_fwd_tail<true, COL_NR>(z, B, Nout, cb * 32, gamma, beta, eps, 1, y, mean, var, momentum, rm, rv, nbt, lds);
}

template <int VW, bool VEC>
__global__ __launch_bounds__(VW * 256) void fc_bwd_bn_col_kernel(
    const float *__restrict__ g, const float *__restrict__ x, const float *__restrict__ W, int B, int K, int Nout,
    int tilesW, float *__restrict__ dW, float *dy, float *__restrict__ db, int zero_db,
    const float *__restrict__ zbn, const float *__restrict__ ybn, const float *__restrict__ gamma,
    const float *__restrict__ mean, const float *__restrict__ var, float eps, float *__restrict__ dz,
    float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ float lds[VW * FC_BWD_LDS_FLOATS];
    const int vw = threadIdx.x >> 8;
    const int kblocks = (K + 31) >> 5;
    float *mine = lds + vw * FC_BWD_LDS_FLOATS;
    if ((int)blockIdx.x < kblocks) {   // column kbk of dx (all row blocks), then the BatchNorm backward of that column
        const int kbk = blockIdx.x;
        fc_bwd_tile<VEC>(g, x, W, B, K, Nout, false, vw * kblocks + kbk, dW, dy, db, zero_db, mine, vw * 32 < B);
        __syncthreads();
        bn1d_bwd_tail<true, COL_NR>(dy, zbn, ybn, B, K, kbk * 32, gamma, mean, var, eps, 1, dz, dgamma, dbeta, lds);
    } else {                           // dW (+ db) tiles, VW per workgroup
        const int t = ((int)blockIdx.x - kblocks) * VW + vw;
        fc_bwd_tile<VEC>(g, x, W, B, K, Nout, true, t < tilesW ? t : 0, dW, dy, db, zero_db, mine, t < tilesW);
    }
}

template <int VW>
int launch_fwd(const float *in, int B, int K, const float *W, const float *bias, int Nout, const float *gamma,
               const float *beta, float eps, float momentum, float *z, float *y, float *mean, float *var, float *rm,
               float *rv, long long *nbt, hipStream_t sm) {
    hipLaunchKernelGGL(fc_bn_col_kernel<VW>, dim3((Nout + 31) / 32), dim3(VW * 256), 0, sm, in, B, K, W, bias, Nout, gamma,
                       beta, eps, momentum, z, y, mean, var, rm, rv, nbt);
    return pngpd_launch_status();
}

template <int VW>
int launch_bwd(const float *g, const float *x, const float *W, int B, int K, int Nout, float *dW, float *dy, float *db,
               int zero_db, const float *zbn, const float *ybn, const float *gamma, const float *mean, const float *var,
               float eps, float *dz, float *dgamma, float *dbeta, hipStream_t sm) {
    const int kblocks = (K + 31) / 32, tilesW = ((Nout + 31) / 32) * kblocks;
    const dim3 grid(kblocks + (tilesW + VW - 1) / VW);
    if ((Nout & 7) == 0)
        hipLaunchKernelGGL((fc_bwd_bn_col_kernel<VW, true>), grid, dim3(VW * 256), 0, sm, g, x, W, B, K, Nout, tilesW, dW,
                           dy, db, zero_db, zbn, ybn, gamma, mean, var, eps, dz, dgamma, dbeta);
    else
        hipLaunchKernelGGL((fc_bwd_bn_col_kernel<VW, false>), grid, dim3(VW * 256), 0, sm, g, x, W, B, K, Nout, tilesW, dW,
                           dy, db, zero_db, zbn, ybn, gamma, mean, var, eps, dz, dgamma, dbeta);
    return pngpd_launch_status();
}

}  // namespace

// The shapes the column kernels cover: B <= 128, the layer on the K-split forward tile (what pngpd_fc_fwd picks: K a
// multiple of 32), the BatchNorm's width a multiple of 32 is NOT required (partial column blocks are masked).
bool pngpd_fc_col_ok(int B, int K) { return B >= 1 && B <= 128 && (K & 31) == 0; }

int pngpd_fc_bn_col_fwd(const float *in, int B, int K, const float *W, const float *bias, int Nout, const float *gamma,
                        const float *beta, float eps, float momentum, float *z, float *y, float *mean, float *var,
                        float *rm, float *rv, long long *nbt, void *stream) {
    if (!in || !W || !bias || !gamma || !beta || !z || !y || !mean || !var || !pngpd_fc_col_ok(B, K) || Nout <= 0 ||
        (rm && !rv))
        return PNGPD_ERR_INVALID_ARG;
    hipStream_t sm = (hipStream_t)stream;
    switch ((B + 31) / 32) {
        case 1: return launch_fwd<1>(in, B, K, W, bias, Nout, gamma, beta, eps, momentum, z, y, mean, var, rm, rv, nbt, sm);
        case 2: return launch_fwd<2>(in, B, K, W, bias, Nout, gamma, beta, eps, momentum, z, y, mean, var, rm, rv, nbt, sm);
        case 3: return launch_fwd<3>(in, B, K, W, bias, Nout, gamma, beta, eps, momentum, z, y, mean, var, rm, rv, nbt, sm);
        default: return launch_fwd<4>(in, B, K, W, bias, Nout, gamma, beta, eps, momentum, z, y, mean, var, rm, rv, nbt, sm);
    }
}

// g (B,Nout) = upstream gradient of y = x W^T + b; x (B,K) = relu(bn(zbn)) = ybn's layer output; dW / db as pngpd_fc_bwd
// (zero_db: the exact-zero bias gradient ahead of a train-mode BatchNorm); dy (B,K) scratch; dz / dgamma / dbeta = the
// backward of the BatchNorm (+ ReLU) that produced x.
int pngpd_fc_bwd_bn_col(const float *g, const float *x, const float *W, int B, int K, int Nout, float *dW, float *dy,
                        float *db, int zero_db, const float *zbn, const float *gamma, const float *mean,
                        const float *var, float eps, float *dz, float *dgamma, float *dbeta, void *stream) {
    if (!g || !x || !W || !dW || !dy || !db || !zbn || !gamma || !mean || !var || !dz || !dgamma || !dbeta || B < 1 ||
        B > 128 || K <= 0 || Nout <= 0)
        return PNGPD_ERR_INVALID_ARG;
    hipStream_t sm = (hipStream_t)stream;
    switch ((B + 31) / 32) {
        case 1: return launch_bwd<1>(g, x, W, B, K, Nout, dW, dy, db, zero_db, zbn, x, gamma, mean, var, eps, dz, dgamma, dbeta, sm);
        case 2: return launch_bwd<2>(g, x, W, B, K, Nout, dW, dy, db, zero_db, zbn, x, gamma, mean, var, eps, dz, dgamma, dbeta, sm);
        case 3: return launch_bwd<3>(g, x, W, B, K, Nout, dW, dy, db, zero_db, zbn, x, gamma, mean, var, eps, dz, dgamma, dbeta, sm);
        default: return launch_bwd<4>(g, x, W, B, K, Nout, dW, dy, db, zero_db, zbn, x, gamma, mean, var, eps, dz, dgamma, dbeta, sm);
    }
}
