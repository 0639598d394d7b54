// libpngpd — inference path: BN fold/pack, fused per-point MLP + max-pool, FC stack.
// Hand-written for gfx950 (MI355X): 64-wide waves, v_mfma_f32_32x32x2_f32 (exact fp32),
// activations staged in LDS, the (B,1024,N) layer-3 activation never touches HBM.
//
// Reference call sites replaced (relative to the reference root):
//   PointNetGPD/model/pointnet.py:29-33   STN3d trunk + MaxPool1d
//   PointNetGPD/model/pointnet.py:140-149 PointNetfeat bmm + trunk + MaxPool1d
//   PointNetGPD/model/pointnet.py:35-43   STN3d FC stack + identity
//   PointNetGPD/model/pointnet.py:191-194 PointNetCls head + log_softmax
#include "pngpd_common.h"

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

// ---------------------------------------------------------------------------------------
// Fused trunk: x (B,3,N) -> pooled partial maxima.
// One workgroup (4 waves) owns cloud b and a contiguous range of 64-point tiles.
// Per tile:  [xs <- x(+T^T)] | layer1 VALU -> h1 (LDS) | layer2 MFMA -> h2 (LDS) |
//            layer3 MFMA, running max per channel in LDS.  Bias/ReLU of layer 3 commute
//            with the max and are applied once at the end.
// ---------------------------------------------------------------------------------------
#define TP 64     // points per tile
#define H1S 68    // h1 row stride in floats (64 + 4 pad: conflict-free ds_read_b128)
#define H2S 132   // h2 row stride in floats (128 + 4 pad)
#define TRUNK_LDS_FLOATS (TP * H1S + TP * H2S + 3 * TP + 1024)

__global__ __launch_bounds__(256, 2) void trunk_infer_kernel(
    const float *__restrict__ x, int N, const float *__restrict__ trans,
    const float *__restrict__ w1, const float *__restrict__ b1,
    const float *__restrict__ w2p, const float *__restrict__ b2,
    const float *__restrict__ w3p, const float *__restrict__ b3,
    int relu_last, int T, int S, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *h1 = smem;              // [TP][H1S]
    float *h2 = h1 + TP * H1S;     // [TP][H2S]
    float *xs = h2 + TP * H2S;     // [3][TP]
    float *rm = xs + 3 * TP;       // [1024] running max of layer-3 pre-bias output

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / S, s = blockIdx.x - b * S;
    const int t0 = (int)(((long)s * T) / S), t1 = (int)(((long)(s + 1) * T) / S);
    const float *xb = x + (size_t)b * 3 * N;

    for (int i = tid; i < 1024; i += 256) rm[i] = -INFINITY;

    float tm[9];
    if (trans) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tm[i] = trans[(size_t)b * 9 + i];
    }

    for (int tile = t0; tile < t1; ++tile) {
        // ---- stage the tile's points (tail: replicate the last point; max is unaffected)
        if (tid < TP) {
            int n = tile * TP + tid;
            n = n < N ? n : N - 1;
            float x0 = xb[n], x1 = xb[N + n], x2 = xb[2 * N + n];
            if (trans) {   // x' = x^T @ trans  (pointnet.py:140-143): x'_c = sum_i x_i * T[i][c]
                float y0 = fmaf(x2, tm[6], fmaf(x1, tm[3], x0 * tm[0]));
                float y1 = fmaf(x2, tm[7], fmaf(x1, tm[4], x0 * tm[1]));
                float y2 = fmaf(x2, tm[8], fmaf(x1, tm[5], x0 * tm[2]));
                x0 = y0; x1 = y1; x2 = y2;
            }
            xs[tid] = x0; xs[TP + tid] = x1; xs[2 * TP + tid] = x2;
        }
        __syncthreads();
        // ---- layer 1 (3 -> 64), VALU: thread = (point p, 16-channel group = wave)
        {
            const int p = lane;
            const float x0 = xs[p], x1 = xs[TP + p], x2 = xs[2 * TP + p];
            float *dst = h1 + p * H1S + wave * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = wave * 16 + g * 4 + e;   // wave-uniform -> scalar loads
                    float a = fmaf(w1[c * 3 + 2], x2, fmaf(w1[c * 3 + 1], x1, fmaf(w1[c * 3], x0, b1[c])));
                    v[e] = fmaxf(a, 0.f);
                }
                *(f32x4 *)(dst + g * 4) = v;
            }
        }
        __syncthreads();
        // ---- layer 2 (64 -> 128), MFMA: wave owns channel block cb = wave, both point blocks
        {
            const int cb = wave;
            const f32x4 *wp = (const f32x4 *)w2p + (size_t)(cb * 8) * 64 + lane;
            f32x16 acc0 = {0}, acc1 = {0};
            const float *a0p = h1 + j * H1S + h * 4;
            const float *a1p = h1 + (32 + j) * H1S + h * 4;
#pragma unroll
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
            const float bias = b2[cb * 32 + j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma_row(r, lane);
                h2[row * H2S + cb * 32 + j] = fmaxf(acc0[r] + bias, 0.f);
                h2[(32 + row) * H2S + cb * 32 + j] = fmaxf(acc1[r] + bias, 0.f);
            }
        }
        __syncthreads();
        // ---- layer 3 (128 -> 1024), MFMA + in-register max over the tile's 64 points
        {
            const float *a0p = h2 + j * H2S + h * 4;
            const float *a1p = h2 + (32 + j) * H2S + h * 4;
#pragma unroll 1
            for (int ci = 0; ci < 8; ++ci) {
                const int cb = wave + 4 * ci;
                const f32x4 *wp = (const f32x4 *)w3p + (size_t)(cb * 16) * 64 + lane;
                f32x4 wf[16];
#pragma unroll
                for (int kb = 0; kb < 16; ++kb) wf[kb] = wp[kb * 64];
                f32x16 acc0 = {0}, acc1 = {0};
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
                float m = fmaxf(acc0[0], acc1[0]);
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, fmaxf(acc0[r], acc1[r]));
                m = fmaxf(m, __shfl_xor(m, 32));   // rows 4*h+... live in the other half-wave
                if (h == 0) rm[cb * 32 + j] = fmaxf(rm[cb * 32 + j], m);
            }
        }
        // no barrier needed here: the next tile's xs/h1 writes do not alias h2/rm, and the
        // barrier before its layer 2 orders them after every wave's layer-3 reads of h2.
    }
    // ---- epilogue: bias (+ReLU) after the max, one coalesced 128-B store per channel block
    if (h == 0) {
        float *o = out + ((size_t)b * S + s) * 1024;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            const int c = (wave + 4 * ci) * 32 + j;
            float v = rm[c] + b3[c];
            if (relu_last) v = fmaxf(v, 0.f);
            o[c] = v;
        }
    }
}

__global__ void pool_reduce_kernel(const float *__restrict__ part, int S, float *__restrict__ out, int total) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over B*1024
    if (idx >= total) return;
    int b = idx >> 10, c = idx & 1023;
    const float *p = part + (size_t)b * S * 1024 + c;
    float m = p[0];
    for (int s = 1; s < S; ++s) m = fmaxf(m, p[(size_t)s * 1024]);
    out[idx] = m;
}

// ---------------------------------------------------------------------------------------
// FC layer: out = epi(in @ W^T + bias).  One wave = 32 samples x 32 outputs, K contracted
// with v_mfma_f32_32x32x2_f32; A (samples) and B (weight rows) fragments are float4 loads
// straight from global (both operands are small and L2-resident).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fc_kernel(const float *__restrict__ in, int B, int K,
                                                 const float *__restrict__ W, const float *__restrict__ bias,
                                                 int Nout, int epi, float *__restrict__ out) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int rb = blockIdx.x, cb = blockIdx.y * 4 + wave;
    if (cb * 32 >= Nout) return;   // wave-uniform; the kernel has no barriers
    int row = rb * 32 + j; row = row < B ? row : B - 1;
    int col = cb * 32 + j; col = col < Nout ? col : Nout - 1;
    const f32x4 *ap = (const f32x4 *)(in + (size_t)row * K) + h;
    const f32x4 *wp = (const f32x4 *)(W + (size_t)col * K) + h;
    f32x16 acc = {0};
    const int KB = K >> 3;
#pragma unroll 8
    for (int kb = 0; kb < KB; ++kb) {
        f32x4 a = ap[kb * 2];
        f32x4 w = wp[kb * 2];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = mfma32(a[t], w[t], acc);
    }
    const int c = cb * 32 + j;
    const bool cvalid = c < Nout;
    const float bv = cvalid ? bias[c] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int orow = rb * 32 + mfma_row(r, lane);
        float v = acc[r] + bv;
        if (epi == PNGPD_EPI_RELU) {
            v = fmaxf(v, 0.f);
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
static int g_trunk_target_blocks = 2048;

static int trunk_splits(int B, int T) {
    int S = (g_trunk_target_blocks + B - 1) / B;
    if (S < 1) S = 1;
    if (S > T) S = T;
    return S;
}

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

int pngpd_set_option(const char *name, int value) {
    if (!name) return PNGPD_ERR_INVALID_ARG;
    if (!strcmp(name, "trunk_target_blocks")) { g_trunk_target_blocks = value > 0 ? value : 1; return PNGPD_OK; }
    return PNGPD_ERR_INVALID_ARG;
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

size_t pngpd_trunk_workspace_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    int T = (N + TP - 1) / TP;
    // sized for the largest split count any option setting can choose (S <= T)
    return (size_t)B * T * 1024 * sizeof(float);
}

int pngpd_trunk_fwd_infer(const float *x, int B, int N, const float *trans,
                          const float *w1, const float *b1, const float *w2p, const float *b2,
                          const float *w3p, const float *b3, int relu_last,
                          float *out_pool, void *workspace, size_t workspace_bytes, void *stream) {
    if (!x || !w1 || !b1 || !w2p || !b2 || !w3p || !b3 || !out_pool || B <= 0 || N <= 0)
        return PNGPD_ERR_INVALID_ARG;
    const int T = (N + TP - 1) / TP;
    const int S = trunk_splits(B, T);
    float *dst = out_pool;
    if (S > 1) {
        if (!workspace || workspace_bytes < (size_t)B * S * 1024 * sizeof(float)) return PNGPD_ERR_WORKSPACE;
        dst = (float *)workspace;
    }
    const size_t lds = TRUNK_LDS_FLOATS * sizeof(float);
    hipLaunchKernelGGL(trunk_infer_kernel, dim3((unsigned)B * S), dim3(256), lds, (hipStream_t)stream,
                       x, N, trans, w1, b1, w2p, b2, w3p, b3, relu_last, T, S, dst);
    int st = pngpd_launch_status();
    if (st != PNGPD_OK) return st;
    if (S > 1) {
        int total = B * 1024;
        hipLaunchKernelGGL(pool_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (const float *)workspace, S, out_pool, total);
        st = pngpd_launch_status();
    }
    return st;
}

int pngpd_fc_fwd(const float *in, int B, int K, const float *W, const float *bias, int Nout,
                 int epilogue, float *out, void *stream) {
    if (!in || !W || !bias || !out || B <= 0 || K <= 0 || Nout <= 0 || (K & 7)) return PNGPD_ERR_INVALID_ARG;
    if (epilogue < PNGPD_EPI_NONE || epilogue > PNGPD_EPI_LOG_SOFTMAX) return PNGPD_ERR_INVALID_ARG;
    if (epilogue == PNGPD_EPI_ADD_IDEN3 && Nout != 9) return PNGPD_ERR_INVALID_ARG;
    if (epilogue == PNGPD_EPI_LOG_SOFTMAX && Nout > 32) return PNGPD_ERR_INVALID_ARG;
    dim3 grid((B + 31) / 32, (Nout + 127) / 128);
    hipLaunchKernelGGL(fc_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, B, K, W, bias, Nout, epilogue, out);
    return pngpd_launch_status();
}

}  // extern "C"
