// libpngpd — the GPD-baseline row of SURVEY.md §8f-4 (the comparator model of the paper and its preprocessing):
//   * projection images of the in-gripper points   PointNetGPD/model/dataset.py:88-198 (project_pc / cal_projection)
//   * depth-map registration + back-projection     PointNetGPD/ycb_cloud_generate.py:60-184
//   * GPDClassifier forward                        PointNetGPD/model/gpd.py:5-31
// All of it is integer / fp64 index work or small convolutions: HBM- and latency-bound, no matrix cores (the two FC
// layers of the classifier go through pngpd_fc_fwd).  fp64 expressions keep the reference's association and use no
// FMA contraction, so voxel indices, pixel indices and depths are bit-identical to numpy's.
#include "pngpd_common.h"

// The fp64 geometry in this file must round exactly like numpy's (separate multiply and add): hipcc's default
// -ffp-contract=fast-honor-pragmas would otherwise fuse a*b + c into one FMA (the pn_dmul/pn_dadd helpers are
// plain operators in HIP's headers and do not prevent it).
#pragma clang fp contract(off)

// ---------------------------------------------------------------------------------------
// Projection images (dataset.py:139-198).  One workgroup per (grasp, projection order).
//   voxel = floor(coord / res + size/2) per axis of the order, res = gripper_width / (size - margin)
//   per (x, y) pixel the voxel with the LARGEST z index wins (np.unique sorts, fancy assignment keeps the last);
//   of that voxel the first `vpn` points in input order are kept; normal = float32 sequential sum / count (f64);
//   occupancy = count / max count.  Points whose normal has a NaN component are skipped (dataset.py:97-101).
// LDS: zmax[S*S] int | cnt[S*S] int | sum[S*S][3] float | chunk keys u16[1024] | chunk normals float[1024][3]
// The per-pixel accumulation is done by the pixel's OWNER thread (pixel % 256) while all threads walk the chunk in
// input order, which makes the float32 sums sequential and deterministic without any sort.
// ---------------------------------------------------------------------------------------
#define GPD_S 60
#define GPD_NPIX (GPD_S * GPD_S)
#define GPD_CHUNK 1024
#define GPD_PROJ_LDS (GPD_NPIX * 4 * 5 + GPD_CHUNK * 2 + GPD_CHUNK * 12 + 64)

__device__ __forceinline__ double gpd_blk_minmax(double v, bool is_max, double *red) {
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
        const double o = __shfl_xor(v, k);
        v = is_max ? (o > v ? o : v) : (o < v ? o : v);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) r = is_max ? (red[w] > r ? red[w] : r) : (red[w] < r ? red[w] : r);
    return r;
}

__global__ __launch_bounds__(256) void gpd_projection_kernel(
    const double *__restrict__ pts, const double *__restrict__ nrm, const int *__restrict__ offsets,
    const double *__restrict__ widths, int chann, int margin, int vpn, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int *zmax = (int *)smem_raw;
    int *cnt = zmax + GPD_NPIX;
    float *sum = (float *)(cnt + GPD_NPIX);                      // [NPIX][3]
    unsigned short *ckey = (unsigned short *)(sum + GPD_NPIX * 3);   // [CHUNK]
    float *cnrm = (float *)(ckey + GPD_CHUNK);                   // [CHUNK][3]
    __shared__ double red[4];
    __shared__ int ired[4];
    const int nproj = chann == 3 ? 1 : 3;
    const int g = blockIdx.x / nproj, pj = blockIdx.x - g * nproj;
    const int tid = threadIdx.x;
    // ORDERS = (0,1,2), (1,2,0), (0,2,1)   dataset.py:104,110,113
    const int o0 = pj == 1 ? 1 : 0, o1 = pj == 0 ? 1 : 2, o2 = pj == 0 ? 2 : (pj == 1 ? 0 : 1);
    const int p0 = offsets[g], M = offsets[g + 1] - p0;
    const double *P = pts + (size_t)p0 * 3, *Nn = nrm + (size_t)p0 * 3;
    double *o = out + (size_t)g * GPD_NPIX * chann;
    const int cbase = chann == 3 ? 0 : 4 * pj;                   // channel of the occupancy plane (12-channel layout)
    auto valid = [&](int i) { return Nn[3 * i] == Nn[3 * i] && Nn[3 * i + 1] == Nn[3 * i + 1] && Nn[3 * i + 2] == Nn[3 * i + 2]; };
    // ---- extents of the first two axes over the kept points (dataset.py:144-153)
    double mx0 = -INFINITY, mn0 = INFINITY, mx1 = -INFINITY, mn1 = INFINITY;
    for (int i = tid; i < M; i += 256) {
        if (!valid(i)) continue;
        const double a = P[3 * i + o0], b = P[3 * i + o1];
        mx0 = a > mx0 ? a : mx0; mn0 = a < mn0 ? a : mn0; mx1 = b > mx1 ? b : mx1; mn1 = b < mn1 ? b : mn1;
    }
    mx0 = gpd_blk_minmax(mx0, true, red); mn0 = gpd_blk_minmax(mn0, false, red);
    mx1 = gpd_blk_minmax(mx1, true, red); mn1 = gpd_blk_minmax(mn1, false, red);
    const double e0 = pn_dsub(mx0, mn0), e1 = pn_dsub(mx1, mn1);
    const double tmp = e0 > e1 ? e0 : e1;
    const bool empty = !(tmp > 0.0);                             // no kept point, or a single location: zero images
    for (int i = tid; i < GPD_NPIX; i += 256) { zmax[i] = INT_MIN; cnt[i] = 0; sum[3 * i] = 0.f; sum[3 * i + 1] = 0.f; sum[3 * i + 2] = 0.f; }
    __syncthreads();
    const double res = pn_ddiv(widths[g], (double)(GPD_S - margin));
    const double half = (double)GPD_S / 2.0;
    auto vox = [&](int i, int &ix, int &iy, int &iz) {
        ix = (int)floor(pn_dadd(pn_ddiv(P[3 * i + o0], res), half));
        iy = (int)floor(pn_dadd(pn_ddiv(P[3 * i + o1], res), half));
        iz = (int)floor(pn_dadd(pn_ddiv(P[3 * i + o2], res), half));
    };
    if (!empty) {
        // ---- pass 1: the winning (largest) z index of every pixel
        for (int i = tid; i < M; i += 256) {
            if (!valid(i)) continue;
            int ix, iy, iz; vox(i, ix, iy, iz);
            if (ix >= 0 && ix < GPD_S && iy >= 0 && iy < GPD_S) atomicMax(&zmax[ix * GPD_S + iy], iz);
        }
        __syncthreads();
        // ---- pass 2: in input order, chunk by chunk; the owner thread of a pixel accumulates its winner voxel
        for (int base = 0; base < M; base += GPD_CHUNK) {
            const int n = (M - base) < GPD_CHUNK ? (M - base) : GPD_CHUNK;
            for (int j = tid; j < n; j += 256) {
                const int i = base + j;
                unsigned short key = 0xFFFFu;
                if (valid(i)) {
                    int ix, iy, iz; vox(i, ix, iy, iz);
                    if (ix >= 0 && ix < GPD_S && iy >= 0 && iy < GPD_S && iz == zmax[ix * GPD_S + iy])
                        key = (unsigned short)(ix * GPD_S + iy);
                    cnrm[3 * j] = (float)Nn[3 * i]; cnrm[3 * j + 1] = (float)Nn[3 * i + 1]; cnrm[3 * j + 2] = (float)Nn[3 * i + 2];
                }
                ckey[j] = key;
            }
            __syncthreads();
            for (int j = 0; j < n; ++j) {
                const unsigned k = ckey[j];                       // one address for the whole wave: LDS broadcast
                if (k != 0xFFFFu && (k & 255u) == (unsigned)tid && cnt[k] < vpn) {
                    sum[3 * k] += cnrm[3 * j]; sum[3 * k + 1] += cnrm[3 * j + 1]; sum[3 * k + 2] += cnrm[3 * j + 2];
                    cnt[k] += 1;
                }
            }
            __syncthreads();
        }
    }
    // ---- occupancy normalisation and output
    int cm = 0;
    for (int i = tid; i < GPD_NPIX; i += 256) cm = cnt[i] > cm ? cnt[i] : cm;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) { const int ov = __shfl_xor(cm, k); cm = ov > cm ? ov : cm; }
    if ((tid & 63) == 0) ired[tid >> 6] = cm;
    __syncthreads();
    cm = max(max(ired[0], ired[1]), max(ired[2], ired[3]));
    for (int i = tid; i < GPD_NPIX; i += 256) {
        const int c = cnt[i];
        double *oi = o + (size_t)i * chann + cbase;
        if (chann == 3) {
            oi[0] = c ? pn_ddiv((double)sum[3 * i], (double)c) : 0.0;
            oi[1] = c ? pn_ddiv((double)sum[3 * i + 1], (double)c) : 0.0;
            oi[2] = c ? pn_ddiv((double)sum[3 * i + 2], (double)c) : 0.0;
        } else {
            oi[0] = cm ? pn_ddiv((double)c, (double)cm) : 0.0;
            oi[1] = c ? pn_ddiv((double)sum[3 * i], (double)c) : 0.0;
            oi[2] = c ? pn_ddiv((double)sum[3 * i + 1], (double)c) : 0.0;
            oi[3] = c ? pn_ddiv((double)sum[3 * i + 2], (double)c) : 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------
// registerDepthMap (ycb_cloud_generate.py:60-121): every depth pixel is carried into the RGB camera; per RGB pixel
// the LARGEST transformed depth survives (the reference's `>`), which makes the loop order-independent: one thread
// per depth pixel and an atomicMax on the bit pattern of the (positive) double.  cam = depthK fx,fy,cx,cy |
// rgbK fx,fy,cx,cy | H (3x4 row-major) = 20 doubles.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void depth_register_kernel(const double *__restrict__ depth, int hd, int wd,
                                                             const double *__restrict__ cam, int hr, int wr,
                                                             unsigned long long *__restrict__ reg) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= hd * wd) return;
    const double d = depth[idx];
    if (d == 0.0) return;
    const int v = idx / wd, u = idx - v * wd;
    const double inv_fx = pn_ddiv(1.0, cam[0]), inv_fy = pn_ddiv(1.0, cam[1]);
    const double x = pn_dmul(pn_dmul(pn_dsub((double)u, cam[2]), d), inv_fx);
    const double y = pn_dmul(pn_dmul(pn_dsub((double)v, cam[3]), d), inv_fy);
    const double *H = cam + 8;
    auto row = [&](int r) {
        return pn_dadd(pn_dadd(pn_dadd(pn_dmul(H[4 * r], x), pn_dmul(H[4 * r + 1], y)), pn_dmul(H[4 * r + 2], d)),
                         H[4 * r + 3]);
    };
    const double X = row(0), Y = row(1), Z = row(2);
    const double inv = pn_ddiv(1.0, Z);
    const double uu = pn_dadd(pn_dadd(pn_dmul(pn_dmul(cam[4], X), inv), cam[6]), 0.5);
    const double vv = pn_dadd(pn_dadd(pn_dmul(pn_dmul(cam[5], Y), inv), cam[7]), 0.5);
    if (!(uu > -1.0e9 && uu < 1.0e9 && vv > -1.0e9 && vv < 1.0e9)) return;   // int() of these would be out of any image
    const int ur = (int)uu, vr = (int)vv;                                     // truncation toward zero, like int()
    if (ur < 0 || ur >= wr || vr < 0 || vr >= hr) return;
    if (Z > 0.0) atomicMax(&reg[(size_t)vr * wr + ur], (unsigned long long)__double_as_longlong(Z));
}

// registeredDepthMapToPointCloud, organized=False (ycb_cloud_generate.py:124-184): ordered compaction of the pixels
// with depth > 0 (row-major) + two rigid transforms.  Pass 1 counts per 256-pixel block, the host-side entry scans
// the (<= a few thousand) block counts with one small kernel, pass 2 emits.
__global__ __launch_bounds__(256) void depth_count_kernel(const double *__restrict__ depth, int n, int *__restrict__ bcnt) {
    __shared__ int w[4];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool ok = idx < n && depth[idx] > 0.0;
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) bcnt[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

__global__ __launch_bounds__(1024) void block_scan_kernel(int *__restrict__ bcnt, int nb, int *__restrict__ total) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    int s = 0;
    for (int i = tid * per; i < nb && i < (tid + 1) * per; ++i) s += bcnt[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;   // exclusive prefix of this thread's range
    for (int i = tid * per; i < nb && i < (tid + 1) * per; ++i) { const int c = bcnt[i]; bcnt[i] = run; run += c; }
    if (tid == 1023) *total = part[1023];
}

__global__ __launch_bounds__(256) void depth_emit_kernel(const double *__restrict__ depth, int h, int w_,
                                                         const double *__restrict__ cam /* rgbK fx,fy,cx,cy | A 3x4 | O 3x4 */,
                                                         const unsigned char *__restrict__ rgb,
                                                         const int *__restrict__ boff, double *__restrict__ xyz,
                                                         unsigned char *__restrict__ rgb_out) {
    __shared__ int wc[4];
    const int idx = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = h * w_;
    const double d = idx < n ? depth[idx] : 0.0;
    const bool ok = d > 0.0;
    const unsigned long long m = __ballot(ok);
    if (lane == 0) wc[wave] = __popcll(m);
    __syncthreads();
    if (!ok) return;
    int pos = boff[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
    for (int q = 0; q < wave; ++q) pos += wc[q];
    const int v = idx / w_, u = idx - v * w_;
    const double x = pn_dmul(pn_dmul(pn_dsub((double)u, cam[2]), d), pn_ddiv(1.0, cam[0]));
    const double y = pn_dmul(pn_dmul(pn_dsub((double)v, cam[3]), d), pn_ddiv(1.0, cam[1]));
    auto rigid = [&](const double *A, int r, double a, double b, double c) {
        return pn_dadd(pn_dadd(pn_dadd(pn_dmul(A[4 * r], a), pn_dmul(A[4 * r + 1], b)), pn_dmul(A[4 * r + 2], c)),
                         A[4 * r + 3]);
    };
    const double *A = cam + 4, *O = cam + 16;
    const double x1 = rigid(A, 0, x, y, d), y1 = rigid(A, 1, x, y, d), z1 = rigid(A, 2, x, y, d);
    xyz[(size_t)pos * 3] = rigid(O, 0, x1, y1, z1);
    xyz[(size_t)pos * 3 + 1] = rigid(O, 1, x1, y1, z1);
    xyz[(size_t)pos * 3 + 2] = rigid(O, 2, x1, y1, z1);
    if (rgb && rgb_out) {
        rgb_out[(size_t)pos * 3] = rgb[(size_t)idx * 3];
        rgb_out[(size_t)pos * 3 + 1] = rgb[(size_t)idx * 3 + 1];
        rgb_out[(size_t)pos * 3 + 2] = rgb[(size_t)idx * 3 + 2];
    }
}

// ---------------------------------------------------------------------------------------
// GPDClassifier (gpd.py:5-31): Conv2d(C,20,5) -> MaxPool2d(2) -> Conv2d(20,50,5) -> MaxPool2d(2) -> fc1 -> ReLU -> fc2
// -> log_softmax.  One kernel = 5x5 valid convolution + bias + 2x2 max-pool (note: the reference has NO activation
// between the convolutions).  Workgroup = (sample, group of OCG output channels); the input planes are staged in LDS
// in chunks of CCH channels, the group's weights in LDS (broadcast reads); a thread owns one pooled pixel and
// accumulates the 2x2 window's four convolution results for its OCG channels from a 6x6 input window.
// The two FC layers run on pngpd_fc_fwd (MFMA).
// ---------------------------------------------------------------------------------------
template <int OCG, int CCH, bool ARG>
__global__ __launch_bounds__(256) void conv5_pool2_kernel(const float *__restrict__ in, int Cin, int Hin,
                                                          const float *__restrict__ W, const float *__restrict__ bias,
                                                          int Cout, float *__restrict__ out,
                                                          unsigned char *__restrict__ arg) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Hc = Hin - 4, Hp = Hc / 2;                       // conv output / pooled output size (square images)
    float *plane = sm;                                          // [CCH][Hin*Hin]
    const int ngroups = (Cout + OCG - 1) / OCG;
    const int b = blockIdx.x / ngroups, og = blockIdx.x - b * ngroups;
    const int tid = threadIdx.x;
    const float *inb = in + (size_t)b * Cin * Hin * Hin;
    {
        const int pp = blockIdx.y * 256 + tid;                  // blockIdx.y: round of 256 pooled pixels
        const bool act = pp < Hp * Hp;
        const int py = act ? pp / Hp : 0, px = act ? pp - (pp / Hp) * Hp : 0;
        float acc[OCG][4];
#pragma unroll
        for (int q = 0; q < OCG; ++q) { acc[q][0] = 0.f; acc[q][1] = 0.f; acc[q][2] = 0.f; acc[q][3] = 0.f; }
        for (int c0 = 0; c0 < Cin; c0 += CCH) {
            const int nc = (Cin - c0) < CCH ? (Cin - c0) : CCH;
            __syncthreads();
            for (int i = tid; i < nc * Hin * Hin; i += 256) plane[i] = inb[(size_t)c0 * Hin * Hin + i];
            __syncthreads();
            if (act) {
                for (int c = 0; c < nc; ++c) {
                    float win[6][6];
                    const float *pl = plane + c * Hin * Hin + (2 * py) * Hin + 2 * px;
#pragma unroll
                    for (int yy = 0; yy < 6; ++yy)
#pragma unroll
                        for (int xx = 0; xx < 6; xx += 2) {      // 2 * px and Hin are even: 8-byte aligned pairs
                            const float2 v = *(const float2 *)(pl + yy * Hin + xx);
                            win[yy][xx] = v.x; win[yy][xx + 1] = v.y;
                        }
#pragma unroll
                    for (int q = 0; q < OCG; ++q) {
                        // the filter of (output channel, input plane) is the same for every thread: read at a uniform
                        // global address it arrives through the scalar unit, no LDS broadcast per multiply-add
                        const int ocq = (og * OCG + q) < Cout ? (og * OCG + q) : (Cout - 1);
                        const float *wq = W + ((size_t)ocq * Cin + c0 + c) * 25;
#pragma unroll
                        for (int ky = 0; ky < 5; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 5; ++kx) {
                                const float wv = wq[ky * 5 + kx];
                                acc[q][0] = fmaf(win[ky][kx], wv, acc[q][0]);
                                acc[q][1] = fmaf(win[ky][kx + 1], wv, acc[q][1]);
                                acc[q][2] = fmaf(win[ky + 1][kx], wv, acc[q][2]);
                                acc[q][3] = fmaf(win[ky + 1][kx + 1], wv, acc[q][3]);
                            }
                    }
                }
            }
        }
        if (act) {
#pragma unroll
            for (int q = 0; q < OCG; ++q) {
                const int oc = og * OCG + q;
                if (oc < Cout) {
                    const float m = fmaxf(fmaxf(acc[q][0], acc[q][1]), fmaxf(acc[q][2], acc[q][3])) + bias[oc];
                    out[(((size_t)b * Cout + oc) * Hp + py) * Hp + px] = m;
                    if (ARG) {      // which of the window's four positions the maximum came from: the FIRST in row-major
                                    // order on a tie, like ATen's max_pool2d scan (strict >) — where its backward routes
                        int code = 0;
                        float mv = acc[q][0];
                        if (acc[q][1] > mv) { mv = acc[q][1]; code = 1; }
                        if (acc[q][2] > mv) { mv = acc[q][2]; code = 2; }
                        if (acc[q][3] > mv) { code = 3; }
                        arg[(((size_t)b * Cout + oc) * Hp + py) * Hp + px] = (unsigned char)code;
                    }
                }
            }
        }
    }
}

extern "C" {

int pngpd_gpd_projection(const double *points, const double *normals, const int *offsets, const double *widths,
                         int G, int chann, int project_size, int margin, int voxel_point_num, double *out,
                         void *stream) {
    if (!points || !normals || !offsets || !widths || !out || G <= 0 || (chann != 3 && chann != 12) ||
        voxel_point_num <= 0 || margin < 0 || margin >= project_size)
        return PNGPD_ERR_INVALID_ARG;
    if (project_size != GPD_S) return PNGPD_ERR_UNSUPPORTED;   // the reference itself only supports 60 (dataset.py:221)
    int st = pngpd_allow_lds((const void *)gpd_projection_kernel, GPD_PROJ_LDS);
    if (st != PNGPD_OK) return st;
    hipLaunchKernelGGL(gpd_projection_kernel, dim3((unsigned)G * (chann == 3 ? 1 : 3)), dim3(256), GPD_PROJ_LDS,
                       (hipStream_t)stream, points, normals, offsets, widths, chann, margin, voxel_point_num, out);
    return pngpd_launch_status();
}

int pngpd_depth_register(const double *depth, int hd, int wd, const double *cam20, int hr, int wr,
                         double *registered, void *stream) {
    if (!depth || !cam20 || !registered || hd <= 0 || wd <= 0 || hr <= 0 || wr <= 0) return PNGPD_ERR_INVALID_ARG;
    hipError_t e = hipMemsetAsync(registered, 0, (size_t)hr * wr * sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) return PNGPD_ERR_HIP + (int)e;
    hipLaunchKernelGGL(depth_register_kernel, dim3((hd * wd + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       depth, hd, wd, cam20, hr, wr, (unsigned long long *)registered);
    return pngpd_launch_status();
}

size_t pngpd_depth_cloud_workspace_bytes(int h, int w) {
    if (h <= 0 || w <= 0) return 0;
    return ((size_t)((h * w + 255) / 256) + 1) * sizeof(int);
}

int pngpd_depth_to_cloud(const double *depth, int h, int w, const double *cam28, const unsigned char *rgb,
                         double *xyz, unsigned char *rgb_out, int *count, void *workspace, size_t workspace_bytes,
                         void *stream) {
    if (!depth || !cam28 || !xyz || !count || !workspace || h <= 0 || w <= 0) return PNGPD_ERR_INVALID_ARG;
    if (workspace_bytes < pngpd_depth_cloud_workspace_bytes(h, w)) return PNGPD_ERR_WORKSPACE;
    const int n = h * w, nb = (n + 255) / 256;
    int *bcnt = (int *)workspace;
    hipLaunchKernelGGL(depth_count_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, depth, n, bcnt);
    int st = pngpd_launch_status();
    if (st != PNGPD_OK) return st;
    hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, bcnt, nb, count);
    st = pngpd_launch_status();
    if (st != PNGPD_OK) return st;
    hipLaunchKernelGGL(depth_emit_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, depth, h, w, cam28, rgb, bcnt,
                       xyz, rgb_out);
    return pngpd_launch_status();
}

static int conv5_pool2_launch(const float *in, int B, int Cin, int Hin, const float *W, const float *bias, int Cout,
                              float *out, unsigned char *arg, void *stream) {
    if (!in || !W || !bias || !out || B <= 0 || Cin <= 0 || Cout <= 0 || Hin < 6 || ((Hin - 4) & 1))
        return PNGPD_ERR_INVALID_ARG;
    constexpr int OCG = 5, CCH = 4;
    const size_t lds = (size_t)CCH * Hin * Hin * sizeof(float);
    if (lds > 150 * 1024) return PNGPD_ERR_UNSUPPORTED;
    int st = pngpd_allow_lds(arg ? (const void *)conv5_pool2_kernel<OCG, CCH, true>
                                 : (const void *)conv5_pool2_kernel<OCG, CCH, false>, lds);
    if (st != PNGPD_OK) return st;
    const int ngroups = (Cout + OCG - 1) / OCG;
    const int Hp = (Hin - 4) / 2;
    const dim3 grid((unsigned)B * ngroups, (Hp * Hp + 255) / 256);
    if (arg)
        hipLaunchKernelGGL((conv5_pool2_kernel<OCG, CCH, true>), grid, dim3(256), lds, (hipStream_t)stream,
                           in, Cin, Hin, W, bias, Cout, out, arg);
    else
        hipLaunchKernelGGL((conv5_pool2_kernel<OCG, CCH, false>), grid, dim3(256), lds, (hipStream_t)stream,
                           in, Cin, Hin, W, bias, Cout, out, arg);
    return pngpd_launch_status();
}

int pngpd_conv5_pool2(const float *in, int B, int Cin, int Hin, const float *W, const float *bias, int Cout,
                      float *out, void *stream) {
    return conv5_pool2_launch(in, B, Cin, Hin, W, bias, Cout, out, nullptr, stream);
}

int pngpd_conv5_pool2_arg(const float *in, int B, int Cin, int Hin, const float *W, const float *bias, int Cout,
                          float *out, unsigned char *arg, void *stream) {
    if (!arg) return PNGPD_ERR_INVALID_ARG;
    return conv5_pool2_launch(in, B, Cin, Hin, W, bias, Cout, out, arg, stream);
}

}  // extern "C"
