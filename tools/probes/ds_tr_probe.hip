// Probe of ds_read_b64_tr_b16 (gfx950): every lane reads 8 bytes at an address of its choosing; what does it get?
// LDS is filled with halfword i = i; lane L reads at byte address base(L).  Printed: per lane the 4 halfwords received.
// usage: ./ds_tr_probe <row_stride_halfwords> <mode>
//   mode 0: lane L (of a 16-lane group g = L/16, i = L%16) reads row (i/4), columns 4*(i%4).. of a [4][16] block g
//   mode 1: lane reads row (i%4), columns 4*(i/4)..
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int stride, int mode, unsigned short *out) {
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int L = threadIdx.x, g = L / 16, i = L % 16;
    int row, col;
    if (mode == 0) { row = i / 4; col = 4 * (i % 4); } else { row = i % 4; col = 4 * (i / 4); }
    const int off = g * 1024 + row * stride + col;      // halfwords
    typedef __attribute__((address_space(3))) bf16x4 lds_v4;
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4 *)(lds + off));
    unsigned short r[4];
    __builtin_memcpy(r, &v, 8);
    for (int e = 0; e < 4; ++e) out[L * 4 + e] = r[e];
}
int main(int argc, char **argv) {
    int stride = argc > 1 ? atoi(argv[1]) : 16, mode = argc > 2 ? atoi(argv[2]) : 0;
    unsigned short *d, h[256];
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, stride, mode, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("stride %d mode %d  (value = g*1024 + row*stride + col)\n", stride, mode);
    for (int L = 0; L < 32; ++L) {
        printf("lane %2d:", L);
        for (int e = 0; e < 4; ++e) { int v = h[L * 4 + e] % 1024; printf("  (r%d,c%2d)", v / stride, v % stride); }
        printf("\n");
    }
    return 0;
}
