// Probe: operand layout of v_mfma_f32_32x32x16_bf16 on gfx950 (asymmetric integer matrices, exact in bf16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short f2bf(float x) { unsigned u = __float_as_uint(x); return (unsigned short)(u >> 16); }

__global__ void probe(const float *A /*32x16*/, const float *B /*16x32*/, float *C /*32x32*/) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int t = 0; t < 8; ++t) {
        unsigned short av = f2bf(A[i * 16 + h * 8 + t]);       // hypothesis: A[i = l&31][k = 8*(l>>5) + t]
        unsigned short bv = f2bf(B[(h * 8 + t) * 32 + i]);     //             B[k = 8*(l>>5) + t][j = l&31]
        a[t] = __builtin_bit_cast(__bf16, av);
        b[t] = __builtin_bit_cast(__bf16, bv);
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        C[row * 32 + i] = c[r];
    }
}
int main() {
    std::vector<float> A(32 * 16), B(16 * 32), C(32 * 32), R(32 * 32, 0.f);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 3 + k * 7) % 11 - 5);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 5 + j * 2 + 1) % 13 - 6);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j]; R[i * 32 + j] = s; }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) if (C[i] != R[i]) ++bad;
    printf("mfma_f32_32x32x16_bf16 layout probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    return bad != 0;
}
