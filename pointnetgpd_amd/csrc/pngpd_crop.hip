// libpngpd — batched in-gripper crop + resample (the upstream of the PointNet scorer).
//
// Reference call sites replaced (relative to the reference root):
//   PointNetGPD/model/dataset.py:53-76     BaseGraspDataset.collect_pc  (training-style box)
//   dex-net/apps/kinect2grasp.py:186-233   check_collision_square, use_dataset_py branch
//   dex-net/apps/kinect2grasp.py:238-258   collect_pc (python loop over grasps)
//   PointNetGPD/model/dataset.py:438-444, kinect2grasp.py:473-478  np.random.choice resampling
//
// All geometry is fp64 like the reference's numpy (clouds are fp64 .npy files for training,
// fp32 ROS clouds promoted to fp64 by the subtraction with an fp64 centre at inference), with
// STRICT inequalities on all six faces.  A grasp frame is 18 doubles:
//   origin[3], M[9] (rows = approach, binormal, minor), lo[3], hi[3]
//   y = M (p - origin);  keep  lo < y < hi  component-wise.
#include "pngpd_common.h"

struct Frame {
    double o[3], m[9], lo[3], hi[3];
};

__device__ __forceinline__ void load_frame(const double *__restrict__ f, Frame &F) {
#pragma unroll
    for (int i = 0; i < 3; ++i) F.o[i] = f[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) F.m[i] = f[3 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { F.lo[i] = f[12 + i]; F.hi[i] = f[15 + i]; }
}

template <bool F64>
__device__ __forceinline__ void load_point(const void *__restrict__ cloud, int p, double &x, double &y, double &z) {
    if (F64) {
        const double *c = (const double *)cloud + (size_t)p * 3;
        x = c[0]; y = c[1]; z = c[2];
    } else {
        const float *c = (const float *)cloud + (size_t)p * 3;
        x = (double)c[0]; y = (double)c[1]; z = (double)c[2];
    }
}

__device__ __forceinline__ void to_frame(const Frame &F, double x, double y, double z, double &a, double &b, double &c) {
    const double dx = x - F.o[0], dy = y - F.o[1], dz = z - F.o[2];
    // same association as a row-times-column dot product: ((m0*dx + m1*dy) + m2*dz), no FMA
    a = __dadd_rn(__dadd_rn(__dmul_rn(F.m[0], dx), __dmul_rn(F.m[1], dy)), __dmul_rn(F.m[2], dz));
    b = __dadd_rn(__dadd_rn(__dmul_rn(F.m[3], dx), __dmul_rn(F.m[4], dy)), __dmul_rn(F.m[5], dz));
    c = __dadd_rn(__dadd_rn(__dmul_rn(F.m[6], dx), __dmul_rn(F.m[7], dy)), __dmul_rn(F.m[8], dz));
}

// One workgroup per grasp: count the in-box points and write their indices in ascending order
// (== np.where(...)[0]) — ordered stream compaction by wave ballots + a 4-entry LDS prefix.
template <bool F64>
__global__ __launch_bounds__(256) void crop_count_compact_kernel(
    const void *__restrict__ cloud, int P, const double *__restrict__ frames, int max_keep,
    int *__restrict__ counts, int *__restrict__ idx) {
    __shared__ int wcnt[4];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Frame F;
    load_frame(frames + (size_t)g * 18, F);
    int running = 0;
    int *out = idx + (size_t)g * max_keep;
    for (int base = 0; base < P; base += 256) {
        const int p = base + tid;
        bool in = false;
        if (p < P) {
            double x, y, z, a, b, c;
            load_point<F64>(cloud, p, x, y, z);
            to_frame(F, x, y, z, a, b, c);
            in = (a > F.lo[0]) && (a < F.hi[0]) && (b > F.lo[1]) && (b < F.hi[1]) && (c > F.lo[2]) && (c < F.hi[2]);
        }
        const unsigned long long mask = __ballot(in);
        if (lane == 0) wcnt[wave] = __popcll(mask);
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = wcnt[w]; if (w < wave) woff += c; total += c; }
        if (in) {
            const int pos = running + woff + __popcll(mask & ((1ull << lane) - 1ull));
            if (pos < max_keep) out[pos] = p;
        }
        running += total;
        __syncthreads();
    }
    if (tid == 0) counts[g] = running;
}

// Counter-based RNG (splitmix64 finaliser) — reproducible per (seed, grasp, draw).
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// One workgroup per grasp: pick N of the m = min(count, max_keep) kept points and write them in the
// hand frame as (3,N) fp32.  Rule of the reference: WITHOUT replacement iff m > N (mode 0, dataset.py:439)
// or m >= N (mode 1, kinect2grasp.py:474); otherwise WITH replacement.  sel != NULL injects the draw:
// sel[g][n] in [0, m) is the rank of the kept point to take (tests / bit-reproducible pipelines).
template <bool F64>
__global__ __launch_bounds__(256) void crop_resample_kernel(
    const void *__restrict__ cloud, const double *__restrict__ frames, const int *__restrict__ counts,
    const int *__restrict__ idx, int max_keep, int N, int mode, int min_points, unsigned long long seed,
    const int *__restrict__ sel, float *__restrict__ out, unsigned char *__restrict__ valid) {
    extern __shared__ int perm[];   // [max_keep] only used for the without-replacement draw
    const int g = blockIdx.x, tid = threadIdx.x;
    const int cnt = counts[g];
    const int m = cnt < max_keep ? cnt : max_keep;
    float *o = out + (size_t)g * 3 * N;
    const bool ok = cnt >= min_points && m > 0;
    if (tid == 0) valid[g] = ok ? 1 : 0;
    if (!ok) {
        for (int i = tid; i < 3 * N; i += 256) o[i] = 0.f;
        return;
    }
    Frame F;
    load_frame(frames + (size_t)g * 18, F);
    const int *gi = idx + (size_t)g * max_keep;
    const bool without = (mode == 0) ? (m > N) : (m >= N);
    if (!sel && without) {
        for (int i = tid; i < m; i += 256) perm[i] = i;
        __syncthreads();
        if (tid == 0) {   // partial Fisher-Yates: the first N entries become a uniform N-subset
            for (int i = 0; i < N; ++i) {
                const unsigned long long r = mix64(seed ^ mix64(((unsigned long long)g << 32) | (unsigned)i));
                const int j = i + (int)(r % (unsigned long long)(m - i));
                const int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
            }
        }
        __syncthreads();
    }
    for (int n = tid; n < N; n += 256) {
        int r;
        if (sel) {
            r = sel[(size_t)g * N + n];
            r = r < 0 ? 0 : (r >= m ? m - 1 : r);
        } else if (without) {
            r = perm[n];
        } else {
            r = (int)(mix64(seed ^ mix64(((unsigned long long)g << 32) | (unsigned)n)) % (unsigned long long)m);
        }
        double x, y, z, a, b, c;
        load_point<F64>(cloud, gi[r], x, y, z);
        to_frame(F, x, y, z, a, b, c);
        o[n] = (float)a; o[N + n] = (float)b; o[2 * N + n] = (float)c;
    }
}

extern "C" {

int pngpd_crop_count_compact(const void *cloud, int cloud_is_f64, int P, const double *frames, int G,
                             int max_keep, int *counts, int *idx, void *stream) {
    if (!cloud || !frames || !counts || !idx || P <= 0 || G <= 0 || max_keep <= 0) return PNGPD_ERR_INVALID_ARG;
    if (cloud_is_f64)
        hipLaunchKernelGGL(crop_count_compact_kernel<true>, dim3(G), dim3(256), 0, (hipStream_t)stream,
                           cloud, P, frames, max_keep, counts, idx);
    else
        hipLaunchKernelGGL(crop_count_compact_kernel<false>, dim3(G), dim3(256), 0, (hipStream_t)stream,
                           cloud, P, frames, max_keep, counts, idx);
    return pngpd_launch_status();
}

int pngpd_crop_resample(const void *cloud, int cloud_is_f64, const double *frames, int G, const int *counts,
                        const int *idx, int max_keep, int N, int mode, int min_points,
                        unsigned long long seed, const int *sel, float *out, unsigned char *valid, void *stream) {
    if (!cloud || !frames || !counts || !idx || !out || !valid || G <= 0 || max_keep <= 0 || N <= 0 ||
        (mode != 0 && mode != 1))
        return PNGPD_ERR_INVALID_ARG;
    const size_t lds = (size_t)max_keep * sizeof(int);
    if (lds > 150 * 1024) return PNGPD_ERR_UNSUPPORTED;
    if (cloud_is_f64) {
        if (lds > 48 * 1024)
            hipFuncSetAttribute((const void *)crop_resample_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(crop_resample_kernel<true>, dim3(G), dim3(256), lds, (hipStream_t)stream,
                           cloud, frames, counts, idx, max_keep, N, mode, min_points, seed, sel, out, valid);
    } else {
        if (lds > 48 * 1024)
            hipFuncSetAttribute((const void *)crop_resample_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(crop_resample_kernel<false>, dim3(G), dim3(256), lds, (hipStream_t)stream,
                           cloud, frames, counts, idx, max_keep, N, mode, min_points, seed, sel, out, valid);
    }
    return pngpd_launch_status();
}

}  // extern "C"
