// libpngpd — BN fold / weight layout, and the FC layer kernel (inference + every FC GEMM of training).
//
// Reference call sites replaced (relative to the reference root):
//   PointNetGPD/model/pointnet.py:35-43   STN3d FC stack + identity
//   PointNetGPD/model/pointnet.py:191-194 PointNetCls head + log_softmax
//   eval-mode bn(conv(x)) / bn(fc(x)) pairs  :29-31,35-36,144-147,191-192  (folded into the weights)
#include "pngpd_common.h"
#include "pngpd_bf.h"

// ---------------------------------------------------------------------------------------
// BN fold + weight layout
// ---------------------------------------------------------------------------------------
// MFMA_B layout of a (C,K) weight: the B operand of D[point][chan] += A[point][k]*B[k][chan]
// for channel block cb (32 channels) and k-block kb (8 k-values) is one float4 per lane:
//   lane = h*32 + j  holds  W[cb*32 + j][kb*8 + h*4 + t], t = 0..3
// so a wave fetches a whole (32 x 8) fragment with ONE coalesced 1-KiB global_load_dwordx4,
// and MFMA step t of that k-block contracts k = kb*8 + t (h=0 lanes) and kb*8 + 4 + t (h=1).
__global__ void fold_conv_bn_kernel(const float *__restrict__ W, const float *__restrict__ b,
                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                    const float *__restrict__ mean, const float *__restrict__ var,
                                    float eps, int C, int K, int layout,
                                    float *__restrict__ Wf, float *__restrict__ bf) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * K) return;
    int c = idx / K, k = idx - c * K;
    double s = 1.0;
    if (gamma) s = var ? (double)gamma[c] / sqrt((double)var[c] + (double)eps) : (double)gamma[c];
    float wv = (float)((double)W[idx] * s);
    int o = idx;
    if (layout == PNGPD_LAYOUT_MFMA_B) {
        int cb = c >> 5, j = c & 31, kb = k >> 3, h = (k >> 2) & 1, t = k & 3;
        o = (((cb * (K >> 3) + kb) * 64) + h * 32 + j) * 4 + t;
    }
    Wf[o] = wv;
    if (k == 0) {
        double bb = b ? (double)b[c] : 0.0;
        if (gamma && var) bb = (bb - (double)mean[c]) * s + (double)beta[c];
        bf[c] = (float)bb;
    }
}

// Every layer of a model in one launch (pngpd_fold_model): block -> layer by the prefix table, then the arithmetic of
// fold_conv_bn_kernel, written to up to three layouts.  The bf16 split is split_pack_bf16_kernel's (pngpd_bf.h split2).
struct FoldModelDev {
    pngpd_fold_model_t m;
    int blk0[PNGPD_FOLD_MAX_LAYERS + 1];
};
__global__ __launch_bounds__(256) void fold_model_kernel(const FoldModelDev A) {
    int li = 0;
#pragma unroll 1
    while (li + 1 < A.m.n && (int)blockIdx.x >= A.blk0[li + 1]) ++li;
    const pngpd_fold_layer_t &L = A.m.layer[li];
    const int idx = ((int)blockIdx.x - A.blk0[li]) * 256 + (int)threadIdx.x;
    const int C = L.C, K = L.K;
    if (idx >= C * K) return;
    const int c = idx / K, k = idx - c * K;
    double s = 1.0;
    if (L.gamma) s = L.var ? (double)L.gamma[c] / sqrt((double)L.var[c] + (double)L.eps) : (double)L.gamma[c];
    const float wv = (float)((double)L.W[idx] * s);
    if (L.row) L.row[idx] = wv;
    if (L.mfma) {
        const int cb = c >> 5, j = c & 31, kb = k >> 3, h = (k >> 2) & 1, t = k & 3;
        L.mfma[(((cb * (K >> 3) + kb) * 64) + h * 32 + j) * 4 + t] = wv;
    }
    if (L.x3) {
        u16 hi, lo;
        split2(wv, hi, lo);
        const int cb = c >> 5, j = c & 31, ks = k >> 4, h = (k >> 3) & 1, t = k & 7, KS = K >> 4;
        const size_t base = ((size_t)(cb * KS + ks) * 2) * 64 * 8 + (size_t)(h * 32 + j) * 8 + t;
        u16 *out = (u16 *)L.x3;
        out[base] = hi;
        out[base + 64 * 8] = lo;
    }
    if (k == 0) {
        double bb = L.b ? (double)L.b[c] : 0.0;
        if (L.gamma && L.var) bb = (bb - (double)L.mean[c]) * s + (double)L.beta[c];
        L.bf[c] = (float)bb;
    }
}

// ---------------------------------------------------------------------------------------
// FC layer: out = epi(in @ W^T + bias).  One wave = 32 samples x 32 outputs, K contracted
// with v_mfma_f32_32x32x2_f32; A (samples) and B (weight rows) fragments are float4 loads
// straight from global (both operands are small and L2-resident).
// ---------------------------------------------------------------------------------------
// KSPLIT (up to 2048 output tiles): the workgroup owns ONE 32x32 tile and its four waves each contract a quarter of K
// (the K = 1024 MFMA chain of one wave is 14 us long); partial tiles meet in LDS.
template <bool KSPLIT>
__global__ __launch_bounds__(256) void fc_kernel(const float *__restrict__ in, int B, int K,
                                                 const float *__restrict__ W, const float *__restrict__ bias,
                                                 int Nout, int epi, float *__restrict__ out) {
    __shared__ float red[KSPLIT ? 3 * 16 * 64 : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int rb = blockIdx.x, cb = KSPLIT ? (int)blockIdx.y : (int)blockIdx.y * 4 + wave;
    if (!KSPLIT && cb * 32 >= Nout) return;   // wave-uniform; this variant has no barriers
    int row = rb * 32 + j; row = row < B ? row : B - 1;
    int col = cb * 32 + j; col = col < Nout ? col : Nout - 1;
    const f32x4 *ap = (const f32x4 *)(in + (size_t)row * K) + h;
    const f32x4 *wp = (const f32x4 *)(W + (size_t)col * K) + h;
    f32x16 acc = {0};
    const int KBall = K >> 3;
    const int KB = KSPLIT ? (wave + 1) * (KBall >> 2) : KBall;
    int kb = KSPLIT ? wave * (KBall >> 2) : 0;
    for (; kb + 4 <= KB; kb += 4) {   // 8 loads in flight per lane
        f32x4 a[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = ap[(kb + u) * 2]; w[u] = wp[(kb + u) * 2]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = mfma32(a[u][t], w[u][t], acc);
    }
    for (; kb < KB; ++kb) {
        f32x4 a = ap[kb * 2];
        f32x4 w = wp[kb * 2];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = mfma32(a[t], w[t], acc);
    }
    if (!KSPLIT && (K & 7) == 4) {   // K = 8 m + 4 (GPDClassifier's fc2, K = 500): the last half block — lanes of the upper
                                     // half (k index 1 of each MFMA) contribute zeros
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, w = {0.f, 0.f, 0.f, 0.f};
        if (h == 0) { a = ap[KBall * 2]; w = wp[KBall * 2]; }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = mfma32(a[t], w[t], acc);
    }
    if (KSPLIT) {
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[r] += red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
    }
    const int c = cb * 32 + j;
    const bool cvalid = c < Nout;
    const float bv = cvalid ? bias[c] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int orow = rb * 32 + mfma_row(r, lane);
        float v = acc[r] + bv;
        if (epi == PNGPD_EPI_RELU) {
            v = (v < 0.f) ? 0.f : v;   // NaN-propagating like F.relu (fmaxf would turn a NaN into 0)
        } else if (epi == PNGPD_EPI_ADD_IDEN3) {
            if (c == 0 || c == 4 || c == 8) v += 1.0f;
        } else if (epi == PNGPD_EPI_LOG_SOFTMAX) {
            float vv = cvalid ? v : -INFINITY;
            float mx = vv;
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
            float e = cvalid ? expf(vv - mx) : 0.f;
            float sum = e;
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) sum += __shfl_xor(sum, m);
            v = vv - mx - logf(sum);
        }
        if (cvalid && orow < B) out[(size_t)orow * Nout + c] = v;
    }
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

int pngpd_abi_version(void) { return PNGPD_ABI_VERSION; }

const char *pngpd_strerror(int code) {
    switch (code) {
        case PNGPD_OK: return "ok";
        case PNGPD_ERR_INVALID_ARG: return "invalid argument";
        case PNGPD_ERR_WORKSPACE: return "workspace too small";
        case PNGPD_ERR_UNSUPPORTED: return "unsupported configuration";
        default: break;
    }
    if (code >= PNGPD_ERR_HIP) return hipGetErrorString((hipError_t)(code - PNGPD_ERR_HIP));
    return "unknown error";
}

int pngpd_fold_conv_bn(const float *W, const float *b, const float *gamma, const float *beta,
                       const float *mean, const float *var, float eps, int C, int K, int layout,
                       float *Wf, float *bf, void *stream) {
    if (!W || !Wf || !bf || C <= 0 || K <= 0) return PNGPD_ERR_INVALID_ARG;
    if (gamma && var && (!beta || !mean)) return PNGPD_ERR_INVALID_ARG;
    if (layout == PNGPD_LAYOUT_MFMA_B) {
        if ((C & 31) || (K & 7)) return PNGPD_ERR_INVALID_ARG;
    } else if (layout != PNGPD_LAYOUT_ROWMAJOR) {
        return PNGPD_ERR_INVALID_ARG;
    }
    int total = C * K;
    hipLaunchKernelGGL(fold_conv_bn_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       W, b, gamma, beta, mean, var, eps, C, K, layout, Wf, bf);
    return pngpd_launch_status();
}

int pngpd_fold_model(const pngpd_fold_model_t *m, void *stream) {
    if (!m || m->n <= 0 || m->n > PNGPD_FOLD_MAX_LAYERS) return PNGPD_ERR_INVALID_ARG;
    FoldModelDev A;
    A.m = *m;
    int blk = 0;
    for (int i = 0; i < m->n; ++i) {
        const pngpd_fold_layer_t &L = m->layer[i];
        if (!L.W || !L.bf || L.C <= 0 || L.K <= 0 || (!L.row && !L.mfma && !L.x3)) return PNGPD_ERR_INVALID_ARG;
        if (L.gamma && L.var && (!L.beta || !L.mean)) return PNGPD_ERR_INVALID_ARG;
        if (L.mfma && ((L.C & 31) || (L.K & 7))) return PNGPD_ERR_INVALID_ARG;
        if (L.x3 && ((L.C & 31) || (L.K & 15))) return PNGPD_ERR_INVALID_ARG;
        A.blk0[i] = blk;
        blk += (L.C * L.K + 255) / 256;
    }
    for (int i = m->n; i <= PNGPD_FOLD_MAX_LAYERS; ++i) A.blk0[i] = blk;
    hipLaunchKernelGGL(fold_model_kernel, dim3(blk), dim3(256), 0, (hipStream_t)stream, A);
    return pngpd_launch_status();
}

int pngpd_fc_fwd(const float *in, int B, int K, const float *W, const float *bias, int Nout,
                 int epilogue, float *out, void *stream) {
    if (!in || !W || !bias || !out || B <= 0 || K <= 0 || Nout <= 0 || (K & 3)) return PNGPD_ERR_INVALID_ARG;
    if (epilogue < PNGPD_EPI_NONE || epilogue > PNGPD_EPI_LOG_SOFTMAX) return PNGPD_ERR_INVALID_ARG;
    if (epilogue == PNGPD_EPI_ADD_IDEN3 && Nout != 9) return PNGPD_ERR_INVALID_ARG;
    if (epilogue == PNGPD_EPI_LOG_SOFTMAX && Nout > 32) return PNGPD_ERR_INVALID_ARG;
    // Few output tiles (the FC stacks of this model up to B ~ 4096: 512 tiles for 1024x512): one tile per WORKGROUP, K
    // split over its four waves — 4x the workgroups and a quarter of the dependent-MFMA chain per wave.  (One tile per
    // wave left half the CUs idle at B = 1024: 27 us per layer.)  Many tiles: one tile per wave, no LDS round trip.
    const long tiles = (long)((B + 31) / 32) * ((Nout + 31) / 32);
    if (tiles <= 2048 && (K & 31) == 0) {
        dim3 grid((B + 31) / 32, (Nout + 31) / 32);
        hipLaunchKernelGGL(fc_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, B, K, W, bias, Nout,
                           epilogue, out);
    } else {
        dim3 grid((B + 31) / 32, (Nout + 127) / 128);
        hipLaunchKernelGGL(fc_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, B, K, W, bias, Nout,
                           epilogue, out);
    }
    return pngpd_launch_status();
}

}  // extern "C"
