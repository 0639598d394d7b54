// Probe (kernel experiments only, not product): do VALU instructions overlap fp32 / bf16 MFMAs on one SIMD of gfx950?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define MF32(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#define MBF(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ab), "v"(bb));
#if VOP == 0
#define VF(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#elif VOP == 1
#define VF(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(a));
#elif VOP == 2
#define VF(r) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(a));
#elif VOP == 3
#define VF(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(a));
#elif VOP == 4
#define VF(r) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(r), "v"(a) : "vcc");
#elif VOP == 5
#define VF(r) asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(a));
#elif VOP == 6
#define VF(r) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#elif VOP == 7
#define VF(r) asm volatile("s_nop 3" ::);
#elif VOP == 8
#define VF(r) asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(ldsaddr));
#elif VOP == 9
#define VF(r) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p##r) : "v"(pa));
#elif VOP == 10
#define VF(r) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b));
#elif VOP == 11
#define VF(r) asm volatile("v_bfi_b32 %0, 31, %1, %0" : "+v"(r) : "v"(a));
#elif VOP == 12
#define VF(r) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p##r) : "v"(pa));
#elif VOP == 13
#define VF(r) asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(sm) : "v"(r), "v"(a));
#elif VOP == 14
#define VF(r) asm volatile("v_or_b32 %0, 5, %0" : "+v"(r));
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: every wave: NM MFMAs then NV VALU per iteration (same wave).  MODE 1: waves 0-3 MFMA only, waves 4-7 VALU only.
template <int BF, int NV, int SPLIT>
__global__ __launch_bounds__(512) void k(float *out, int iters, unsigned long long *cyc) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8}, bb = {1, 1, 1, 1, 1, 1, 1, 1};
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    f32x2 pa = {a, b}, pv0 = pa, pv1 = pa, pv2 = pa, pv3 = pa, pv4 = pa, pv5 = pa, pv6 = pa, pv7 = pa; unsigned long long sm = 0;
    const int wave = threadIdx.x >> 6; const int ldsaddr = (threadIdx.x & 63) * 4; (void)ldsaddr;
    const bool do_m = !SPLIT || wave < 4, do_v = !SPLIT || wave >= 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (do_m) {
                if (BF) { MBF(c0) } else { MF32(c0) }
            }
            if (do_v) {
#pragma unroll
                for (int q = 0; q < NV / 8; ++q) { VF(v0) VF(v1) VF(v2) VF(v3) VF(v4) VF(v5) VF(v6) VF(v7) }
            }
            if (do_m) {
                if (BF) { MBF(c1) } else { MF32(c1) }
            }
            if (do_v) {
#pragma unroll
                for (int q = 0; q < NV / 8; ++q) { VF(v0) VF(v1) VF(v2) VF(v3) VF(v4) VF(v5) VF(v6) VF(v7) }
            }
            if (do_m) {
                if (BF) { MBF(c2) } else { MF32(c2) }
            }
            if (do_v) {
#pragma unroll
                for (int q = 0; q < NV / 8; ++q) { VF(v0) VF(v1) VF(v2) VF(v3) VF(v4) VF(v5) VF(v6) VF(v7) }
            }
            if (do_m) {
                if (BF) { MBF(c3) } else { MF32(c3) }
            }
            if (do_v) {
#pragma unroll
                for (int q = 0; q < NV / 8; ++q) { VF(v0) VF(v1) VF(v2) VF(v3) VF(v4) VF(v5) VF(v6) VF(v7) }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + pv0[0] + pv1[1] + pv2[0] + pv3[0] + pv4[0] + pv5[0] + pv6[0] + pv7[0] + (float)sm;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int BF, int NV, int SPLIT>
void run(const char *name, int threads) {
    float *out; unsigned long long *cyc, h[8];
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    hipMemset(cyc, 0, 64);
    const int iters = 2000;
    k<BF, NV, SPLIT><<<256, threads>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<BF, NV, SPLIT><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const double nm = iters * 8.0;
    printf("%-44s thr %3d  %7.3f ms | s_memtime per MFMA-slot: w0 %6.1f", name, threads, ms, h[0] / nm);
    if (threads > 256) printf("  w4 %6.1f", h[4] / nm);
    printf("   (NV %d per MFMA slot)\n", NV);
    hipFree(out); hipFree(cyc);
}

int main() {
    const char *names[] = {"v_fma_f32", "v_cndmask_b32", "v_max_f32", "v_add_u32", "v_cmp_gt_f32 (vcc)", "v_mov_b32", "v_max3_f32",
                           "s_nop 3", "ds_read_b32", "v_pk_fma_f32", "v_min3_u32", "v_bfi_b32", "v_pk_add_f32",
                           "v_cmp_gt_f32 (sgpr)", "v_or_b32"};
    printf("== the interleaved instruction (\"op\") is %s; columns: wall time of the launch | cycles per matrix instruction slot\n", names[VOP]);
    run<0, 0, 0>("fp32 mfma only, 1 wave/SIMD", 256);
    run<0, 8, 0>("fp32 mfma + 8 op same wave, 1 wave/SIMD", 256);
    run<0, 16, 0>("fp32 mfma + 16 op same wave, 1 wave/SIMD", 256);
    run<0, 8, 0>("fp32 mfma + 8 op same wave, 2 waves/SIMD", 512);
    run<1, 8, 0>("bf16 mfma + 8 op same wave, 1 wave/SIMD", 256);
    run<1, 16, 0>("bf16 mfma + 16 op same wave, 1 wave/SIMD", 256);
    return 0;
}
