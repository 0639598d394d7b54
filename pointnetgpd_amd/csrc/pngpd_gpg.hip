// libpngpd — device half of the GPG grasp-candidate sampler, the upstream of the crop.
//
// Reference call sites replaced (dex-net/src/dexnet/grasping/grasp_sampler.py):
//   :1471-1485  r-ball neighbourhood of the sample point (<= 100 nearest within r) and M = sum n n^T
//   :336-393    check_collision_square — the cloud against one box of the hand model in the grasp frame
//   :405-421    check_collide          — bottom plate + both fingers
// which the reference evaluates one numpy call at a time: 19 rotations x 21 offsets x 2..4 boxes per sample
// point (:1524-1565) and up to 25 push-in steps x (3 + 1 + 3) boxes per surviving pose (:1576-1629).
// Here every pose of every sample point is one thread of ONE launch and all boxes are tested on the same
// transformed point.  Geometry is fp64 with strict inequalities and no FMA contraction, like pngpd_crop.hip.
#include "pngpd_common.h"

// The fp64 geometry in this file must round exactly like numpy's (separate multiply and add): hipcc's default
// -ffp-contract=fast-honor-pragmas would otherwise fuse a*b + c into one FMA (the pn_dmul/pn_dadd helpers are
// plain operators in HIP's headers and do not prevent it).
#pragma clang fp contract(off)

template <bool F64>
__device__ __forceinline__ void gpg_load_point(const void *__restrict__ cloud, int p, double &x, double &y, double &z) {
    if (F64) {
        const double *c = (const double *)cloud + (size_t)p * 3;
        x = c[0]; y = c[1]; z = c[2];
    } else {
        const float *c = (const float *)cloud + (size_t)p * 3;
        x = (double)c[0]; y = (double)c[1]; z = (double)c[2];
    }
}

__device__ __forceinline__ double gpg_dist2(double x, double y, double z, double qx, double qy, double qz) {
    const double dx = x - qx, dy = y - qy, dz = z - qz;
    return pn_dadd(pn_dadd(pn_dmul(dx, dx), pn_dmul(dy, dy)), pn_dmul(dz, dz));
}

// Block-wide sum of an int over 256 threads; every thread gets the total.
__device__ __forceinline__ int block_sum_int(int v, int *sh) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// One workgroup per sample point.  Selects the (at most) max_nn nearest cloud points with d^2 < r^2 — ties at
// the cut broken towards the lower index, like a stable sort by distance — and accumulates
// M = sum_{selected, d^2 != 0} n^ n^^T with n^ = n/|n| (n left as is when |n| == 0).
// The k-th smallest squared distance is found by bisection on the bit pattern of the (non-negative) doubles:
// 64 counting passes over an L2-resident cloud, no sort and no per-point storage.
template <bool F64>
__global__ __launch_bounds__(256) void gpg_normal_moments_kernel(
    const void *__restrict__ cloud, const double *__restrict__ normals, int P, const double *__restrict__ queries,
    double r2, int max_nn, double *__restrict__ M_out, int *__restrict__ nsel_out) {
    __shared__ int shi[4];
    __shared__ double shd[4 * 6];
    const int s = blockIdx.x, tid = threadIdx.x;
    const double qx = queries[s * 3 + 0], qy = queries[s * 3 + 1], qz = queries[s * 3 + 2];

    auto count_le_bits = [&](unsigned long long T) {   // #points with d2 < r2 and bits(d2) <= T
        int c = 0;
        for (int p = tid; p < P; p += 256) {
            double x, y, z;
            gpg_load_point<F64>(cloud, p, x, y, z);
            const double d2 = gpg_dist2(x, y, z, qx, qy, qz);
            c += (d2 < r2 && (unsigned long long)__double_as_longlong(d2) <= T) ? 1 : 0;
        }
        return block_sum_int(c, shi);
    };

    const unsigned long long r2bits = (unsigned long long)__double_as_longlong(r2);
    const int in_ball = count_le_bits(r2bits);
    unsigned long long T = r2bits;     // selection: d2 < r2 && bits(d2) <= T [&& tie rule]
    int tie_keep = 0x7fffffff;         // among bits(d2) == T keep those with index <= tie_keep
    if (in_ball > max_nn) {
        unsigned long long lo = 0, hi = r2bits;          // smallest T with count(<= T) >= max_nn
        while (lo < hi) {
            const unsigned long long mid = lo + ((hi - lo) >> 1);
            if (count_le_bits(mid) >= max_nn) hi = mid; else lo = mid + 1;
        }
        T = lo;
        const int n_le = count_le_bits(T);
        if (n_le > max_nn) {                             // ties at the cut: keep the lowest indices
            const int n_lt = T ? count_le_bits(T - 1) : 0;
            const int need = max_nn - n_lt;
            int ilo = 0, ihi = P - 1;                    // smallest I with #(bits == T, idx <= I) >= need
            while (ilo < ihi) {
                const int imid = ilo + ((ihi - ilo) >> 1);
                int c = 0;
                for (int p = tid; p <= imid; p += 256) {
                    double x, y, z;
                    gpg_load_point<F64>(cloud, p, x, y, z);
                    const double d2 = gpg_dist2(x, y, z, qx, qy, qz);
                    c += (d2 < r2 && (unsigned long long)__double_as_longlong(d2) == T) ? 1 : 0;
                }
                if (block_sum_int(c, shi) >= need) ihi = imid; else ilo = imid + 1;
            }
            tie_keep = ilo;
        }
    }

    double m[6] = {0, 0, 0, 0, 0, 0};
    int nsel = 0;
    for (int p = tid; p < P; p += 256) {
        double x, y, z;
        gpg_load_point<F64>(cloud, p, x, y, z);
        const double d2 = gpg_dist2(x, y, z, qx, qy, qz);
        const unsigned long long b = (unsigned long long)__double_as_longlong(d2);
        const bool sel = d2 < r2 && (b < T || (b == T && p <= tie_keep));
        if (!sel) continue;
        ++nsel;
        if (d2 == 0.0) continue;                          // :1477 skips the sample point itself
        double nx = normals[(size_t)p * 3], ny = normals[(size_t)p * 3 + 1], nz = normals[(size_t)p * 3 + 2];
        const double nn = sqrt(pn_dadd(pn_dadd(pn_dmul(nx, nx), pn_dmul(ny, ny)), pn_dmul(nz, nz)));
        if (nn != 0.0) { nx /= nn; ny /= nn; nz /= nn; }
        m[0] += pn_dmul(nx, nx); m[1] += pn_dmul(nx, ny); m[2] += pn_dmul(nx, nz);
        m[3] += pn_dmul(ny, ny); m[4] += pn_dmul(ny, nz); m[5] += pn_dmul(nz, nz);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) m[i] += __shfl_xor(m[i], k);
    nsel = block_sum_int(nsel, shi);
    if ((tid & 63) == 0)
#pragma unroll
        for (int i = 0; i < 6; ++i) shd[(tid >> 6) * 6 + i] = m[i];
    __syncthreads();
    if (tid == 0) {
        double t[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) t[i] = (shd[i] + shd[6 + i]) + (shd[12 + i] + shd[18 + i]);
        double *o = M_out + (size_t)s * 9;
        o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
        o[3] = t[1]; o[4] = t[3]; o[5] = t[4];
        o[6] = t[2]; o[7] = t[4]; o[8] = t[5];
        nsel_out[s] = nsel;
    }
}

// One THREAD per hand pose: pose = 12 doubles [centre, approach, binormal, minor] (axes already unit length, as
// check_collision_square :338-343 makes them); boxes = NB x [x_lo, x_hi, y_lo, y_hi, z_lo, z_hi] in the grasp
// frame (:361-377).  The cloud streams through LDS in chunks that every lane reads at the same address
// (broadcast), so a pose's counters never leave its registers; blockIdx.y splits the cloud when there are too
// few poses to fill the chip, partial counts meet with integer atomics (order-independent -> deterministic).
constexpr int GPG_CHUNK = 1024;

template <bool F64, int NB>
__global__ __launch_bounds__(256) void hand_box_counts_kernel(
    const void *__restrict__ cloud, int P, const double *__restrict__ poses, int Q,
    const double *__restrict__ boxes, int *__restrict__ counts) {
    __shared__ double pts[GPG_CHUNK * 3];
    __shared__ double bx[NB * 6];
    const int tid = threadIdx.x;
    const int q = blockIdx.x * 256 + tid;
    const int per = (P + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * per, p1 = min(P, p0 + per);
    if (tid < NB * 6) bx[tid] = boxes[tid];
    double f[12];
    const int qq = q < Q ? q : Q - 1;
#pragma unroll
    for (int i = 0; i < 12; ++i) f[i] = poses[(size_t)qq * 12 + i];
    int cnt[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) cnt[b] = 0;
    for (int base = p0; base < p1; base += GPG_CHUNK) {
        const int n = min(GPG_CHUNK, p1 - base);
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            double x, y, z;
            gpg_load_point<F64>(cloud, base + i, x, y, z);
            pts[i * 3] = x; pts[i * 3 + 1] = y; pts[i * 3 + 2] = z;
        }
        __syncthreads();
        for (int i = 0; i < n; ++i) {
            const double dx = pts[i * 3] - f[0], dy = pts[i * 3 + 1] - f[1], dz = pts[i * 3 + 2] - f[2];
            const double gx = pn_dadd(pn_dadd(pn_dmul(f[3], dx), pn_dmul(f[4], dy)), pn_dmul(f[5], dz));
            const double gy = pn_dadd(pn_dadd(pn_dmul(f[6], dx), pn_dmul(f[7], dy)), pn_dmul(f[8], dz));
            const double gz = pn_dadd(pn_dadd(pn_dmul(f[9], dx), pn_dmul(f[10], dy)), pn_dmul(f[11], dz));
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const bool in = (bx[b * 6] < gx) && (bx[b * 6 + 1] > gx) && (bx[b * 6 + 2] < gy) &&
                                (bx[b * 6 + 3] > gy) && (bx[b * 6 + 4] < gz) && (bx[b * 6 + 5] > gz);
                cnt[b] += in ? 1 : 0;
            }
        }
    }
    if (q < Q) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (gridDim.y == 1) counts[(size_t)q * NB + b] = cnt[b];
            else if (cnt[b]) atomicAdd(&counts[(size_t)q * NB + b], cnt[b]);
        }
    }
}

// Indexed variant for large clouds / many poses.  The cloud is pre-sorted along a Morton curve and cut into
// 64-point chunks with a bounding sphere each (spheres (C,4) f64 = centre, radius; built by the host half from
// torch ops).  One WAVE per pose: the pose lives in wave-uniform registers; in the broad phase lane c tests chunk
// c's sphere against the hand's bounding box in the grasp frame (|row . v| <= |v| for unit rows, so a chunk whose
// centre is farther than r outside a face cannot hold an in-box point) and a ballot turns the 64 verdicts into a
// work list; the narrow phase runs the exact per-point test of hand_box_counts_kernel (same fp64 operations, so
// the counts are identical) with lane = point and accumulates ballot popcounts in scalar registers.  No atomics,
// no LDS; typically 5-15 % of the chunks survive the broad phase.
template <bool F64, int NB>
__global__ __launch_bounds__(256) void hand_box_counts_indexed_kernel(
    const void *__restrict__ cloud, int P, const double *__restrict__ spheres, int C,
    const double *__restrict__ poses, int Q, const double *__restrict__ boxes, int *__restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;                       // wave-uniform; the kernel has no barriers
    double f[12], bx[NB * 6];
#pragma unroll
    for (int i = 0; i < 12; ++i) f[i] = poses[(size_t)q * 12 + i];
#pragma unroll
    for (int i = 0; i < NB * 6; ++i) bx[i] = boxes[i];
    double hlo[3], hhi[3];                     // bounding box of the hand model in the grasp frame
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        hlo[a] = bx[2 * a]; hhi[a] = bx[2 * a + 1];
#pragma unroll
        for (int b = 1; b < NB; ++b) { hlo[a] = fmin(hlo[a], bx[b * 6 + 2 * a]); hhi[a] = fmax(hhi[a], bx[b * 6 + 2 * a + 1]); }
    }
    int cnt[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) cnt[b] = 0;
    for (int cbase = 0; cbase < C; cbase += 64) {
        const int c = cbase + lane;
        bool pass = false;
        if (c < C) {
            const double4 sp = *(const double4 *)(spheres + (size_t)c * 4);
            const double dx = sp.x - f[0], dy = sp.y - f[1], dz = sp.z - f[2];
            const double r = sp.w * (1.0 + 1e-9) + 1e-12;       // conservative against rounding in the test itself
            const double gx = f[3] * dx + f[4] * dy + f[5] * dz;
            const double gy = f[6] * dx + f[7] * dy + f[8] * dz;
            const double gz = f[9] * dx + f[10] * dy + f[11] * dz;
            pass = gx + r > hlo[0] && gx - r < hhi[0] && gy + r > hlo[1] && gy - r < hhi[1] &&
                   gz + r > hlo[2] && gz - r < hhi[2];
        }
        unsigned long long work = __ballot(pass);
        while (work) {
            const int b0 = __ffsll((long long)work) - 1;
            work &= work - 1ull;
            const int p = (cbase + b0) * 64 + lane;
            bool in[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) in[b] = false;
            if (p < P) {
                double x, y, z;
                gpg_load_point<F64>(cloud, p, x, y, z);
                const double dx = x - f[0], dy = y - f[1], dz = z - f[2];
                const double gx = pn_dadd(pn_dadd(pn_dmul(f[3], dx), pn_dmul(f[4], dy)), pn_dmul(f[5], dz));
                const double gy = pn_dadd(pn_dadd(pn_dmul(f[6], dx), pn_dmul(f[7], dy)), pn_dmul(f[8], dz));
                const double gz = pn_dadd(pn_dadd(pn_dmul(f[9], dx), pn_dmul(f[10], dy)), pn_dmul(f[11], dz));
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    in[b] = (bx[b * 6] < gx) && (bx[b * 6 + 1] > gx) && (bx[b * 6 + 2] < gy) &&
                            (bx[b * 6 + 3] > gy) && (bx[b * 6 + 4] < gz) && (bx[b * 6 + 5] > gz);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) cnt[b] += __popcll(__ballot(in[b]));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) counts[(size_t)q * NB + b] = cnt[b];
    }
}

extern "C" {

int pngpd_gpg_normal_moments(const void *cloud, int cloud_is_f64, const double *normals, int P,
                             const double *queries, int K, double radius, int max_nn, double *M_out,
                             int *nsel_out, void *stream) {
    if (!cloud || !normals || !queries || !M_out || !nsel_out || P <= 0 || K <= 0 || max_nn <= 0 || !(radius > 0))
        return PNGPD_ERR_INVALID_ARG;
    const double r2 = radius * radius;
    if (cloud_is_f64)
        hipLaunchKernelGGL(gpg_normal_moments_kernel<true>, dim3(K), dim3(256), 0, (hipStream_t)stream, cloud,
                           normals, P, queries, r2, max_nn, M_out, nsel_out);
    else
        hipLaunchKernelGGL(gpg_normal_moments_kernel<false>, dim3(K), dim3(256), 0, (hipStream_t)stream, cloud,
                           normals, P, queries, r2, max_nn, M_out, nsel_out);
    return pngpd_launch_status();
}

int pngpd_hand_box_counts(const void *cloud, int cloud_is_f64, int P, const double *poses, int Q,
                          const double *boxes, int num_boxes, int *counts, void *stream) {
    if (!cloud || !poses || !boxes || !counts || P <= 0 || Q <= 0) return PNGPD_ERR_INVALID_ARG;
    if (num_boxes != 1 && num_boxes != 4) return PNGPD_ERR_UNSUPPORTED;
    const int qblocks = (Q + 255) / 256;
    int split = (1024 + qblocks - 1) / qblocks;                 // aim at >= ~1024 workgroups
    const int max_split = (P + GPG_CHUNK - 1) / GPG_CHUNK;
    split = split < 1 ? 1 : (split > max_split ? max_split : split);
    if (split > 1) {
        hipError_t e = hipMemsetAsync(counts, 0, (size_t)Q * num_boxes * sizeof(int), (hipStream_t)stream);
        if (e != hipSuccess) return PNGPD_ERR_HIP + (int)e;
    }
    dim3 grid(qblocks, split);
#define LAUNCH(F64, NB)                                                                                       \
    hipLaunchKernelGGL((hand_box_counts_kernel<F64, NB>), grid, dim3(256), 0, (hipStream_t)stream, cloud, P, \
                       poses, Q, boxes, counts)
    if (cloud_is_f64) { if (num_boxes == 4) LAUNCH(true, 4); else LAUNCH(true, 1); }
    else              { if (num_boxes == 4) LAUNCH(false, 4); else LAUNCH(false, 1); }
#undef LAUNCH
    return pngpd_launch_status();
}

int pngpd_hand_box_counts_indexed(const void *cloud_sorted, int cloud_is_f64, int P, const double *spheres, int C,
                                  const double *poses, int Q, const double *boxes, int num_boxes, int *counts,
                                  void *stream) {
    if (!cloud_sorted || !spheres || !poses || !boxes || !counts || P <= 0 || Q <= 0 || C != (P + 63) / 64)
        return PNGPD_ERR_INVALID_ARG;
    if (num_boxes != 1 && num_boxes != 4) return PNGPD_ERR_UNSUPPORTED;
    dim3 grid((Q + 3) / 4);
#define LAUNCH(F64, NB)                                                                                   \
    hipLaunchKernelGGL((hand_box_counts_indexed_kernel<F64, NB>), grid, dim3(256), 0, (hipStream_t)stream, \
                       cloud_sorted, P, spheres, C, poses, Q, boxes, counts)
    if (cloud_is_f64) { if (num_boxes == 4) LAUNCH(true, 4); else LAUNCH(true, 1); }
    else              { if (num_boxes == 4) LAUNCH(false, 4); else LAUNCH(false, 1); }
#undef LAUNCH
    return pngpd_launch_status();
}

}  // extern "C"
