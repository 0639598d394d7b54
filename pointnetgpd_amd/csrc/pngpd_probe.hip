// Matrix-pipe rate probe: a stream of independent MFMAs and nothing else, one or two waves per SIMD on every CU.
// bench.py times it next to the hot kernels so that the roofline block can quote the SUSTAINED matrix rate of the
// very box it ran on beside the nominal peak (MI355X_MICROARCH.md: 157.3 TFLOP/s fp32 at the 2.4 GHz boost clock; a
// CU array streaming fp32 MFMAs settles near 2.16 GHz = ~141 TFLOP/s, which the inference trunk reaches).
// Also the evidence behind two design rules of this library (tools/probes/mfma_valu_overlap.hip has the full
// experiment): a v_mfma_f32_32x32x2_f32 issues every 64 cycles per SIMD whatever the number of resident waves, and
// VALU instructions do NOT overlap it (their issue time adds), while they do overlap v_mfma_f32_32x32x16_bf16.
#include "pngpd_common.h"

typedef short pr_bf16x8 __attribute__((ext_vector_type(8)));

template <int BF>
__global__ __launch_bounds__(512) void mfma_rate_kernel(float *__restrict__ sink, int iters) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f;
    pr_bf16x8 ab, bb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ab[i] = (short)0x3f80; bb[i] = (short)(0x3f80 + (threadIdx.x & 1)); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (BF) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0);
            } else {
                c0 = mfma32(a, b, c0); c1 = mfma32(a, b, c1); c2 = mfma32(a, b, c2); c3 = mfma32(a, b, c3);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 123.456f) sink[blockIdx.x * blockDim.x + threadIdx.x] = s;   // never true: keeps the chains alive
}

extern "C" int pngpd_probe_mfma_rate(int dtype, int waves_per_simd, int iters, float *sink, long long *flops_out,
                                     void *stream) {
    if ((dtype != 0 && dtype != 1) || (waves_per_simd != 1 && waves_per_simd != 2) || iters <= 0 || !sink)
        return PNGPD_ERR_INVALID_ARG;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return PNGPD_ERR_HIP;
    const int threads = 256 * waves_per_simd;
    if (dtype == 1) hipLaunchKernelGGL(mfma_rate_kernel<1>, dim3(cus), dim3(threads), 0, (hipStream_t)stream, sink, iters);
    else hipLaunchKernelGGL(mfma_rate_kernel<0>, dim3(cus), dim3(threads), 0, (hipStream_t)stream, sink, iters);
    if (flops_out) {
        const long long per_mfma = dtype == 1 ? 2LL * 32 * 32 * 16 : 2LL * 32 * 32 * 2;
        *flops_out = (long long)cus * (threads / 64) * (long long)iters * 8 * per_mfma;
    }
    return pngpd_launch_status();
}
