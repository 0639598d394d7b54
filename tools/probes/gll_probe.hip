// global_load_lds_dwordx4 (gfx950): do the 64 x 16 B of one instruction land in LDS in lane order at M0 + offset?  They do.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/gll_probe.hip -o build_probe/gll_probe   (HISTORY.md 9: weight staging through LDS)
#include <hip/hip_runtime.h>
#include <vector>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const f32x4 *g, f32x4 *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *dst = smem + wave * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(g + (wave * 8 + i) * 64 + lane),
                                         (void __attribute__((address_space(3))) *)(dst + i * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += *(const f32x4 *)(dst + i * 1024 + lane * 16);
    out[threadIdx.x] = acc;
}
int main() {
    f32x4 *g, *o; hipMalloc(&g, 8 * 8 * 64 * 16); hipMalloc(&o, 512 * 16);
    std::vector<float> h(8 * 8 * 64 * 4); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 1000);
    hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(512), 65536, 0, g, o);
    std::vector<float> r(512 * 4); hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 512; ++t) for (int e = 0; e < 4; ++e) { float s = 0; for (int i = 0; i < 8; ++i) s += h[(((t >> 6) * 8 + i) * 64 + (t & 63)) * 4 + e]; bad += s != r[t * 4 + e]; }
    printf("global_load_lds b128: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    return bad != 0;
}
