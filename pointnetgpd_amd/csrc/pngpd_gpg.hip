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

// The fp64 geometry in this file must round exactly like numpy's (separate multiply and add): no FMA contraction for
// anything written below, and the pn_d* helpers of pngpd_common.h for the expressions that mirror the reference.
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
// The k-th smallest squared distance is found by a radix select on the bit pattern of the (non-negative) doubles,
// most significant byte first: 8 histogram passes over an L2-resident cloud, no sort and no per-point storage.
template <bool F64>
__global__ __launch_bounds__(256) void gpg_normal_moments_kernel(
    const void *__restrict__ cloud, const double *__restrict__ normals, int P, const double *__restrict__ queries,
    double r2, int max_nn, double *__restrict__ M_out, int *__restrict__ nsel_out) {
    __shared__ int shi[4];
    __shared__ double shd[4 * 6];
    __shared__ int hist[256];
    __shared__ int pick[2];
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
        // the max_nn-th smallest key, most significant byte first: per byte ONE pass over the cloud builds the histogram
        // of that byte among the candidates that match the prefix found so far, a scan of the 256 bins picks the byte —
        // 8 passes where the bisection on the bit pattern took up to 64 (0.58 -> ~0.15 ms per scene at P = 20,000).
        // Integer counts in LDS: order-independent, so the result is exactly the bisection's.
        unsigned long long lo = 0;          // prefix of the key, bytes above `pos` decided
        int remaining = max_nn;
        for (int pos = 7; pos >= 0; --pos) {
            hist[tid] = 0;
            __syncthreads();
            const int sh_hi = 8 * (pos + 1);
            for (int p0 = 0; p0 < P; p0 += 256) {         // whole waves stay in the loop (ballots)
                const int p = p0 + tid;
                double x, y, z;
                gpg_load_point<F64>(cloud, p < P ? p : P - 1, x, y, z);
                const double d2 = gpg_dist2(x, y, z, qx, qy, qz);
                const unsigned long long key = (unsigned long long)__double_as_longlong(d2);
                const bool match = pos == 7 ? true : (key >> sh_hi) == lo;
                const bool cand = p < P && d2 < r2 && match;
                const int digit = (int)((key >> (8 * pos)) & 255ull);
                if (pos == 7) {
                    // sign / exponent byte: every candidate of a wave lands in one or two bins — wave-aggregated, one LDS
                    // atomic per distinct digit and wave instead of 64 colliding ones
                    unsigned long long active = __ballot(cand);
                    while (active) {
                        const int leader = __ffsll((long long)active) - 1;
                        const int dl = __shfl(digit, leader);
                        const unsigned long long same = __ballot(cand && digit == dl);
                        if ((tid & 63) == leader) atomicAdd(&hist[dl], __popcll(same));
                        active &= ~same;
                    }
                } else if (cand) {
                    atomicAdd(&hist[digit], 1);      // mantissa bytes: the digits of a wave are spread over the bins
                }
            }
            __syncthreads();
            if (tid == 0) {
                int cum = 0, d = 0;
                for (; d < 255; ++d) {
                    if (cum + hist[d] >= remaining) break;
                    cum += hist[d];
                }
                pick[0] = d; pick[1] = remaining - cum;
            }
            __syncthreads();
            lo = (lo << 8) | (unsigned long long)pick[0];
            remaining = pick[1];
            __syncthreads();
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
    const double *__restrict__ poses, int Q, const double *__restrict__ boxes, int *__restrict__ counts,
    const int *__restrict__ valid_units, int per_unit) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;                       // wave-uniform; the kernel has no barriers
    // device-side pose count (the sampler's second sweep: only the first *valid_units * per_unit poses exist, the
    // launch is sized for the capacity so that the host never has to read the count back)
    if (valid_units && q >= *valid_units * per_unit) return;
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


// =======================================================================================================
// Selection logic of the sampler on the device (grasp_sampler.py:1524-1650): pose enumeration, the middle admissible
// offset per rotation, the 30-degree rule, push-in poses with the table back-off, the first accepted push-in step and
// the packing of the result.  (The 3x3 eigen-decomposition ahead of it — whose eigenvector signs decide the enumeration
// order — is gpg_frames_kernel, round 6: LAPACK's DGEEV restated.)  Every expression keeps numpy's operation order.
//   prm (doubles): [0] init_bite [1] hand_depth [2] hand_depth*0.5 [3] APPROACH_STEP [4] TABLE_CLEARANCE
//                  [5] hh*0.5 [6] -(hh*0.5) [7] -(ow*0.5) [8] ow*0.5 [9] -fw [10] fw [11] -hh [12] APPROACH_STEP*3 factor
//                  [16..16+R) dtheta (rad)   [48..48+D) lateral offsets dy   [80..80+S) push-in step numbers
// =======================================================================================================
#define GPG_PRM_DTH 16
#define GPG_PRM_DYS 48
#define GPG_PRM_STEPS 80

__device__ __forceinline__ double gpg_norm3(double x, double y, double z) {
    return sqrt(pn_dadd(pn_dadd(pn_dmul(x, x), pn_dmul(y, y)), pn_dmul(z, z)));
}

// frames (L,12) = minor, normal (flipped), major, sample point.  One thread per (l, r): the rotation about `minor`
// by dtheta[r] (rotation_from_quaternion on the un-normalised [dtheta, minor] quaternion, :1503 quirk), the rotated
// approach / binormal, and the D poses of the lateral sweep (:1524-1541).
__global__ __launch_bounds__(256) void gpg_enumerate_kernel(const double *__restrict__ frames, int L, int R, int D,
                                                            const double *__restrict__ prm,
                                                            double *__restrict__ poses, double *__restrict__ ab) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= L * R) return;
    const int l = t / R, r = t - l * R;
    const double *F = frames + (size_t)l * 12;
    const double mi[3] = {F[0], F[1], F[2]}, no[3] = {F[3], F[4], F[5]}, ma[3] = {F[6], F[7], F[8]};
    double q[4] = {mi[0], mi[1], mi[2], prm[GPG_PRM_DTH + r]};
    const double nq = pn_dadd(pn_dadd(pn_dadd(pn_dmul(q[0], q[0]), pn_dmul(q[1], q[1])), pn_dmul(q[2], q[2])), pn_dmul(q[3], q[3]));
    double rot[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    if (!(nq < 2.220446049250313e-16 * 4.0)) {
        const double sc = sqrt(pn_ddiv(2.0, nq));
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = pn_dmul(q[i], sc);
        double o[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[i][j] = pn_dmul(q[i], q[j]);
        rot[0][0] = pn_dsub(pn_dsub(1.0, o[1][1]), o[2][2]); rot[0][1] = pn_dsub(o[0][1], o[2][3]); rot[0][2] = pn_dadd(o[0][2], o[1][3]);
        rot[1][0] = pn_dadd(o[0][1], o[2][3]); rot[1][1] = pn_dsub(pn_dsub(1.0, o[0][0]), o[2][2]); rot[1][2] = pn_dsub(o[1][2], o[0][3]);
        rot[2][0] = pn_dsub(o[0][2], o[1][3]); rot[2][1] = pn_dadd(o[1][2], o[0][3]); rot[2][2] = pn_dsub(pn_dsub(1.0, o[0][0]), o[1][1]);
    }
    double bi[3], ap[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        bi[i] = pn_dadd(pn_dadd(pn_dmul(rot[i][0], ma[0]), pn_dmul(rot[i][1], ma[1])), pn_dmul(rot[i][2], ma[2]));
        ap[i] = pn_dadd(pn_dadd(pn_dmul(rot[i][0], no[0]), pn_dmul(rot[i][1], no[1])), pn_dmul(rot[i][2], no[2]));
    }
    double *o6 = ab + (size_t)t * 6;
#pragma unroll
    for (int i = 0; i < 3; ++i) { o6[i] = ap[i]; o6[3 + i] = bi[i]; }
    const double na = gpg_norm3(ap[0], ap[1], ap[2]), nb = gpg_norm3(bi[0], bi[1], bi[2]), nm = gpg_norm3(mi[0], mi[1], mi[2]);
    const double ib = prm[0];
    for (int d = 0; d < D; ++d) {
        const double dy = prm[GPG_PRM_DYS + d];
        double *po = poses + ((size_t)t * D + d) * 12;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double b0 = pn_dadd(F[9 + i], pn_dmul(bi[i], dy));               // sample point + binormal * dy
            po[i] = pn_dadd(pn_dmul(ib, -ap[i]), b0);                              // init_bite * (-approach) + ...
            po[3 + i] = pn_ddiv(ap[i], na); po[6 + i] = pn_ddiv(bi[i], nb); po[9 + i] = pn_ddiv(mi[i], nm);
        }
    }
}

// One thread per (l, r): the middle admissible lateral offset (:1565-1567) and the 30-degree rule (:1570-1573).
// flag[t] = 1 and dsel[t] = the offset index when the rotation yields a potential grasp.
__global__ __launch_bounds__(256) void gpg_select_kernel(const int *__restrict__ cnt, const double *__restrict__ poses,
                                                         const double *__restrict__ ab, int LR, int D,
                                                         const double *__restrict__ prm, int *__restrict__ flag,
                                                         int *__restrict__ dsel) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= LR) return;
    const int *c = cnt + (size_t)t * D * 4;
    int n_ok = 0;
    for (int d = 0; d < D; ++d) n_ok += (c[d * 4 + 0] > 0 && c[d * 4 + 3] == 0 && c[d * 4 + 1] == 0 && c[d * 4 + 2] == 0) ? 1 : 0;
    int f = 0, ds = 0;
    if (n_ok > 0) {
        const int target = (n_ok + 1) / 2 - 1;                                     // ceil(n_ok / 2) - 1
        int rank = -1;
        for (int d = 0; d < D; ++d) {
            if (c[d * 4 + 0] > 0 && c[d * 4 + 3] == 0 && c[d * 4 + 1] == 0 && c[d * 4 + 2] == 0) {
                if (++rank == target) { ds = d; break; }
            }
        }
        const double p0z = poses[((size_t)t * D + ds) * 12 + 2], paz = ab[(size_t)t * 6 + 2];
        f = pn_dadd(p0z, pn_dmul(paz, prm[1])) < pn_dsub(p0z, prm[2]) ? 1 : 0;     // (p0 + pa*hd).z < p0.z - hd*0.5
    }
    flag[t] = f; dsel[t] = ds;
}

// Exclusive scan of n 0/1 flags (single workgroup): list[pos] = index of the pos-th set flag, *total = their number.
__global__ __launch_bounds__(1024) void gpg_flag_scan_kernel(const int *__restrict__ flag, int n, int *__restrict__ list,
                                                             int *__restrict__ total) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b = tid * per, e = (b + per < n) ? b + per : n;
    int s = 0;
    for (int i = b; i < e; ++i) s += flag[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int i = b; i < e; ++i) if (flag[i]) list[run++] = i;
    if (tid == 1023) *total = part[1023];
}

// hand corners p1..p20 (:287-321) of pose (centre c, approach a, binormal b) with the construction order of the
// reference (see gpg.py::_hand_table); returns the lowest corner (first minimum of z) in low[] and its z.
__device__ __forceinline__ void gpg_lowest_corner(const double *prm, const double c[3], const double a[3],
                                                  const double b[3], double low[3]) {
    double m[3] = {pn_dsub(pn_dmul(a[1], b[2]), pn_dmul(a[2], b[1])), pn_dsub(pn_dmul(a[2], b[0]), pn_dmul(a[0], b[2])),
                   pn_dsub(pn_dmul(a[0], b[1]), pn_dmul(a[1], b[0]))};
    const double nm = gpg_norm3(m[0], m[1], m[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) m[i] = pn_ddiv(m[i], nm);
    double P[21][3];   // P[0] unused; "u" -> 21st/22nd handled through locals
    double u[3], dn[3];
    auto mk = [&](double (&dst)[3], const double (&par)[3], const double *ax, double sc) {
#pragma unroll
        for (int i = 0; i < 3; ++i) dst[i] = pn_dadd(pn_dmul(ax[i], sc), par[i]);
    };
    const double cc[3] = {c[0], c[1], c[2]};
    mk(u, cc, m, prm[5]); mk(dn, cc, m, prm[6]);
    mk(P[5], u, b, prm[7]); mk(P[6], u, b, prm[8]); mk(P[7], dn, b, prm[8]); mk(P[8], dn, b, prm[7]);
    mk(P[1], P[5], a, prm[1]); mk(P[2], P[6], a, prm[1]); mk(P[3], P[7], a, prm[1]); mk(P[4], P[8], a, prm[1]);
    mk(P[9], P[1], b, prm[9]); mk(P[10], P[4], b, prm[9]); mk(P[11], P[5], b, prm[9]); mk(P[12], P[8], b, prm[9]);
    mk(P[13], P[2], b, prm[10]); mk(P[14], P[3], b, prm[10]); mk(P[15], P[6], b, prm[10]); mk(P[16], P[7], b, prm[10]);
    mk(P[17], P[11], a, prm[11]); mk(P[18], P[15], a, prm[11]); mk(P[19], P[16], a, prm[11]); mk(P[20], P[12], a, prm[11]);
    int best = 1;
#pragma unroll
    for (int i = 2; i <= 20; ++i) if (P[i][2] < P[best][2]) best = i;
    low[0] = P[best][0]; low[1] = P[best][1]; low[2] = P[best][2];
}

// One thread per (potential pose i, push-in step s) (:1575-1612): the pose at step s and its backed-off,
// table-corrected twin.  poses2 (Np,S,2,12) — pose (i,s) and its twin are neighbours, so the first 2*Np*S poses of the
// capacity-sized buffer are the valid ones; back / mod (Np,S,3).
__global__ __launch_bounds__(256) void gpg_pushin_kernel(const int *__restrict__ list, const int *__restrict__ total,
                                                         const int *__restrict__ dsel, const double *__restrict__ poses,
                                                         const double *__restrict__ ab, const double *__restrict__ frames,
                                                         int R, int D, int S, const double *__restrict__ prm,
                                                         double *__restrict__ poses2, double *__restrict__ back_o,
                                                         double *__restrict__ mod_o) {
    const int Np = *total;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Np * S) return;
    const int i = t / S, s = t - i * S;
    const int lr = list[i], l = lr / R;
    const double *p0 = poses + ((size_t)lr * D + dsel[lr]) * 12;
    const double *pa = ab + (size_t)lr * 6, *pb = pa + 3, *pm = frames + (size_t)l * 12;
    const double step = prm[GPG_PRM_STEPS + s], h = prm[3];
    double cs[3], bk[3], md[3], low[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        cs[k] = pn_dadd(pn_dmul(pn_dmul(pa[k], step), h), p0[k]);                  // pa * s * step_len + p0
        bk[k] = pn_dadd(cs[k], pn_dmul(pn_dmul(-pa[k], h), 3.0));                  // + (-pa) * step_len * 3
    }
    gpg_lowest_corner(prm, bk, pa, pb, low);
    const double tx = pn_dadd(pn_ddiv(pn_dmul(-low[2], pa[0]), pa[2]), low[0]);
    const double ty = pn_dadd(pn_ddiv(pn_dmul(-low[2], pa[1]), pa[2]), low[1]);
    const double dist = pn_dadd(sqrt(pn_dadd(pn_dadd(pn_dadd(pn_dadd(pn_dmul(low[0], low[0]), pn_dmul(low[1], low[1])),
                                                             pn_dmul(low[2], low[2])), pn_dmul(tx, tx)), pn_dmul(ty, ty))),
                                prm[4]);
    const bool corr = low[2] < prm[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) md[k] = corr ? pn_dsub(bk[k], pn_dmul(pa[k], dist)) : bk[k];
    const double na = gpg_norm3(pa[0], pa[1], pa[2]), nb = gpg_norm3(pb[0], pb[1], pb[2]), nm = gpg_norm3(pm[0], pm[1], pm[2]);
    double *q0 = poses2 + ((size_t)i * S + s) * 24, *q1 = q0 + 12;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool fin = md[k] == md[k] && fabs(md[k]) != INFINITY;
        q0[k] = cs[k]; q1[k] = fin ? md[k] : 1e30;                                 // non-finite -> a pose no point can hit
        q0[3 + k] = q1[3 + k] = pn_ddiv(pa[k], na);
        q0[6 + k] = q1[6 + k] = pn_ddiv(pb[k], nb);
        q0[9 + k] = q1[9 + k] = pn_ddiv(pm[k], nm);
        back_o[((size_t)i * S + s) * 3 + k] = bk[k];
        mod_o[((size_t)i * S + s) * 3 + k] = md[k];
    }
}

// One thread per potential pose: the first accepted push-in step (:1614-1625; the reference's `break` sits inside the
// accept branch).  found[i] / sfirst[i].
__global__ __launch_bounds__(256) void gpg_first_accept_kernel(const int *__restrict__ cnt2, const int *__restrict__ total,
                                                               int S, int cap, int min_open, int *__restrict__ found,
                                                               int *__restrict__ sfirst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    int f = 0, sf = 0;
    if (i < *total) {
        for (int s = 0; s < S; ++s) {
            const int *a = cnt2 + ((size_t)i * S + s) * 8, *b = a + 4;
            const bool hit0 = a[3] > 0 || a[1] > 0 || a[2] > 0, hit1 = b[3] > 0 || b[1] > 0 || b[2] > 0;
            if (hit0 && b[0] > min_open && !hit1) { f = 1; sf = s; break; }
        }
    }
    found[i] = f; sfirst[i] = sf;
}

// Pack: res = [n_found, per-live-sample-point counts (L), grasps (n_found,5,3)] as doubles, in potential-pose order
// (= the reference's output order: by sample point, then by rotation).
__global__ __launch_bounds__(256) void gpg_pack_kernel(const int *__restrict__ olist, const int *__restrict__ ototal,
                                                       const int *__restrict__ list, const int *__restrict__ sfirst,
                                                       const double *__restrict__ ab, const double *__restrict__ frames,
                                                       const double *__restrict__ back, const double *__restrict__ mod,
                                                       int R, int S, int L, double *__restrict__ res) {
    const int n = *ototal;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j == 0) res[0] = (double)n;
    if (j >= n) return;
    const int i = olist[j], lr = list[i], l = lr / R, s = sfirst[i];
    double *o = res + 1 + L + (size_t)j * 15;
    const double *pa = ab + (size_t)lr * 6, *pm = frames + (size_t)l * 12;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = back[((size_t)i * S + s) * 3 + k];
        o[3 + k] = pa[k]; o[6 + k] = pa[3 + k]; o[9 + k] = pm[k];
        o[12 + k] = mod[((size_t)i * S + s) * 3 + k];
    }
    atomicAdd(&res[1 + l], 1.0);      // small-integer counts: exact and order-independent in fp64
}

#include "pngpd_gpg_sweep.h"
#include "pngpd_gpg_moments.h"
#include "pngpd_gpg_pushin.h"

// ---- local frames of the sample points (:1486-1506) ---------------------------------------------------------------------
// One thread per sample point: np.linalg.eig(M) as LAPACK evaluates it, then the reference's frame construction with
// numpy's roundings — all of it in pngpd_gpg_eig3.h, which the CPU tests compile for the host (same source, contraction
// off in both builds: the kernel's output equals the host build's bit for bit).  frames (K,12) = minor, normal, major,
// sample point — the input of gpg_enumerate_kernel.
#include "pngpd_gpg_eig3.h"

__global__ __launch_bounds__(64) void gpg_frames_kernel(const double *__restrict__ M, const double *__restrict__ nat,
                                                         const double *__restrict__ pts, int K,
                                                         double *__restrict__ frames, int *__restrict__ flags) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= K) return;
    double m[9], na[3], pt[3], f[12];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = M[(size_t)s * 9 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { na[i] = nat[(size_t)s * 3 + i]; pt[i] = pts[(size_t)s * 3 + i]; }
    flags[s] = pn_gpg_local_frame(m, na, pt, f);
#pragma unroll
    for (int i = 0; i < 12; ++i) frames[(size_t)s * 12 + i] = f[i];
}

extern "C" {

int pngpd_gpg_frames(const double *M, const double *normals_at, const double *points, int K, double *frames,
                     int *flags, void *stream) {
    if (!M || !normals_at || !points || !frames || !flags || K <= 0) return PNGPD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(gpg_frames_kernel, dim3((K + 63) / 64), dim3(64), 0, (hipStream_t)stream, M, normals_at, points,
                       K, frames, flags);
    return pngpd_launch_status();
}

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

int pngpd_hand_box_counts_indexed_n(const void *cloud_sorted, int cloud_is_f64, int P, const double *spheres, int C,
                                    const double *poses, int Q, const double *boxes, int num_boxes,
                                    const int *valid_units, int per_unit, int *counts, void *stream) {
    if (!cloud_sorted || !spheres || !poses || !boxes || !counts || P <= 0 || Q <= 0 || C != (P + 63) / 64 ||
        (valid_units && per_unit <= 0))
        return PNGPD_ERR_INVALID_ARG;
    if (num_boxes != 1 && num_boxes != 4) return PNGPD_ERR_UNSUPPORTED;
    dim3 grid((Q + 3) / 4);
#define LAUNCH(F64, NB)                                                                                   \
    hipLaunchKernelGGL((hand_box_counts_indexed_kernel<F64, NB>), grid, dim3(256), 0, (hipStream_t)stream, \
                       cloud_sorted, P, spheres, C, poses, Q, boxes, counts, valid_units, per_unit)
    if (cloud_is_f64) { if (num_boxes == 4) LAUNCH(true, 4); else LAUNCH(true, 1); }
    else              { if (num_boxes == 4) LAUNCH(false, 4); else LAUNCH(false, 1); }
#undef LAUNCH
    return pngpd_launch_status();
}

int pngpd_hand_box_counts_indexed(const void *cloud_sorted, int cloud_is_f64, int P, const double *spheres, int C,
                                  const double *poses, int Q, const double *boxes, int num_boxes, int *counts,
                                  void *stream) {
    return pngpd_hand_box_counts_indexed_n(cloud_sorted, cloud_is_f64, P, spheres, C, poses, Q, boxes, num_boxes,
                                           nullptr, 0, counts, stream);
}

/* ---- selection logic on the device (see the kernels above) ---- */
int pngpd_gpg_enumerate(const double *frames, int L, int R, int D, const double *prm, double *poses, double *ab,
                        void *stream) {
    if (!frames || !prm || !poses || !ab || L <= 0 || R <= 0 || R > 32 || D <= 0 || D > 32) return PNGPD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(gpg_enumerate_kernel, dim3((L * R + 255) / 256), dim3(256), 0, (hipStream_t)stream, frames, L, R, D,
                       prm, poses, ab);
    return pngpd_launch_status();
}

int pngpd_gpg_select(const int *counts, const double *poses, const double *ab, int L, int R, int D, const double *prm,
                     int *flag, int *dsel, int *list, int *total, void *stream) {
    if (!counts || !poses || !ab || !prm || !flag || !dsel || !list || !total || L <= 0 || R <= 0 || D <= 0)
        return PNGPD_ERR_INVALID_ARG;
    const int LR = L * R;
    hipLaunchKernelGGL(gpg_select_kernel, dim3((LR + 255) / 256), dim3(256), 0, (hipStream_t)stream, counts, poses, ab, LR,
                       D, prm, flag, dsel);
    int st = pngpd_launch_status();
    if (st != PNGPD_OK) return st;
    hipLaunchKernelGGL(gpg_flag_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, flag, LR, list, total);
    return pngpd_launch_status();
}

int pngpd_gpg_pushin(const int *list, const int *total, const int *dsel, const double *poses, const double *ab,
                     const double *frames, int L, int R, int D, int S, const double *prm, double *poses2, double *back,
                     double *mod, void *stream) {
    if (!list || !total || !dsel || !poses || !ab || !frames || !prm || !poses2 || !back || !mod || L <= 0 || R <= 0 ||
        D <= 0 || S <= 0 || S > 64)
        return PNGPD_ERR_INVALID_ARG;
    const int cap = L * R;      // capacity: every rotation of every sample point may yield a potential grasp
    hipLaunchKernelGGL(gpg_pushin_kernel, dim3(((size_t)cap * S + 255) / 256), dim3(256), 0, (hipStream_t)stream, list,
                       total, dsel, poses, ab, frames, R, D, S, prm, poses2, back, mod);
    return pngpd_launch_status();
}

int pngpd_gpg_finish(const int *counts2, const int *list, const int *total, const double *ab, const double *frames,
                     const double *back, const double *mod, int L, int R, int S, int min_open, int *found,
                     int *sfirst, int *olist, int *ototal, double *res, void *stream) {
    if (!list || !total || !ab || !frames || !back || !mod || !found || !sfirst || !olist || !ototal ||
        !res || L <= 0 || R <= 0 || S <= 0)
        return PNGPD_ERR_INVALID_ARG;
    const int cap = L * R;
    hipError_t e = hipMemsetAsync(res, 0, (size_t)(1 + L) * sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) return PNGPD_ERR_HIP + (int)e;
    int st = PNGPD_OK;
    if (counts2) {      // NULL: found / sfirst were filled by pngpd_gpg_pushin_sweep
        hipLaunchKernelGGL(gpg_first_accept_kernel, dim3((cap + 255) / 256), dim3(256), 0, (hipStream_t)stream, counts2,
                           total, S, cap, min_open, found, sfirst);
        st = pngpd_launch_status();
        if (st != PNGPD_OK) return st;
    }
    hipLaunchKernelGGL(gpg_flag_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, found, cap, olist, ototal);
    st = pngpd_launch_status();
    if (st != PNGPD_OK) return st;
    hipLaunchKernelGGL(gpg_pack_kernel, dim3((cap + 255) / 256), dim3(256), 0, (hipStream_t)stream, olist, ototal, list,
                       sfirst, ab, frames, back, mod, R, S, L, res);
    return pngpd_launch_status();
}

int pngpd_gpg_sweep_select(const void *cloud_sorted, int cloud_is_f64, int P, const double *spheres, int C,
                           const double *poses, const double *ab, int L, int R, int D, const double *boxes,
                           const double *prm, double tol, int *flag, int *dsel, int *list, int *total,
                           unsigned *masks, unsigned long long *stats, void *stream) {
    if (!cloud_sorted || !spheres || !poses || !ab || !boxes || !prm || !flag || !dsel || !list || !total || P <= 0 ||
        L <= 0 || R <= 0 || D <= 0 || D > 32 || C != (P + 63) / 64 || !(tol >= 0.0))
        return PNGPD_ERR_INVALID_ARG;
    const int LR = L * R;
    if (cloud_is_f64)
        hipLaunchKernelGGL(gpg_sweep_select_kernel<true>, dim3((LR + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                           cloud_sorted, P, spheres, C, poses, ab, LR, D, boxes, prm, tol, flag, dsel, masks, stats);
    else
        hipLaunchKernelGGL(gpg_sweep_select_kernel<false>, dim3((LR + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                           cloud_sorted, P, spheres, C, poses, ab, LR, D, boxes, prm, tol, flag, dsel, masks, stats);
    int st = pngpd_launch_status();
    if (st != PNGPD_OK) return st;
    hipLaunchKernelGGL(gpg_flag_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, flag, LR, list, total);
    return pngpd_launch_status();
}

int pngpd_gpg_normal_moments_indexed(const void *cloud_sorted, int cloud_is_f64, const int *order,
                                     const double *normals, int P, const double *spheres, int C,
                                     const double *queries, int K, double radius, int max_nn, double *M_out,
                                     int *nsel_out, void *stream) {
    if (!cloud_sorted || !order || !normals || !spheres || !queries || !M_out || !nsel_out || P <= 0 || K <= 0 ||
        max_nn <= 0 || !(radius > 0) || C != (P + 63) / 64)
        return PNGPD_ERR_INVALID_ARG;
    if (max_nn > GPG_MAXSEL) return PNGPD_ERR_UNSUPPORTED;      // (use pngpd_gpg_normal_moments)
    if (cloud_is_f64)
        hipLaunchKernelGGL(gpg_normal_moments_indexed_kernel<true>, dim3(K), dim3(256), 0, (hipStream_t)stream,
                           cloud_sorted, order, normals, P, spheres, C, queries, radius, max_nn, M_out, nsel_out);
    else
        hipLaunchKernelGGL(gpg_normal_moments_indexed_kernel<false>, dim3(K), dim3(256), 0, (hipStream_t)stream,
                           cloud_sorted, order, normals, P, spheres, C, queries, radius, max_nn, M_out, nsel_out);
    return pngpd_launch_status();
}

int pngpd_gpg_pushin_sweep(const void *cloud_sorted, int cloud_is_f64, int P, const double *spheres, int C,
                           const double *poses2, const int *total, int L, int R, int S, const double *boxes,
                           int min_open, double tol, int *found, int *sfirst, unsigned long long *stats,
                           void *stream) {
    if (!cloud_sorted || !spheres || !poses2 || !total || !boxes || !found || !sfirst || P <= 0 || L <= 0 || R <= 0 ||
        S <= 0 || C != (P + 63) / 64 || !(tol >= 0.0))
        return PNGPD_ERR_INVALID_ARG;
    if (S > 32) return PNGPD_ERR_UNSUPPORTED;                   // (use pngpd_hand_box_counts_indexed_n + pngpd_gpg_finish)
    const int cap = L * R;
    if (cloud_is_f64)
        hipLaunchKernelGGL(gpg_pushin_sweep_kernel<true>, dim3((cap + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                           cloud_sorted, P, spheres, C, poses2, total, cap, S, boxes, min_open, tol, found, sfirst, stats);
    else
        hipLaunchKernelGGL(gpg_pushin_sweep_kernel<false>, dim3((cap + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                           cloud_sorted, P, spheres, C, poses2, total, cap, S, boxes, min_open, tol, found, sfirst, stats);
    return pngpd_launch_status();
}

}  // extern "C"
