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

// The fp64 geometry in this file must round exactly like numpy's (separate multiply and add): hipcc's default
// -ffp-contract=fast-honor-pragmas would otherwise fuse a*b + c into one FMA (the pn_dmul/pn_dadd helpers are
// plain operators in HIP's headers and do not prevent it).
#pragma clang fp contract(off)

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
    a = pn_dadd(pn_dadd(pn_dmul(F.m[0], dx), pn_dmul(F.m[1], dy)), pn_dmul(F.m[2], dz));
    b = pn_dadd(pn_dadd(pn_dmul(F.m[3], dx), pn_dmul(F.m[4], dy)), pn_dmul(F.m[5], dz));
    c = pn_dadd(pn_dadd(pn_dmul(F.m[6], dx), pn_dmul(F.m[7], dy)), pn_dmul(F.m[8], dz));
}

// One workgroup per grasp: count the in-box points and write their indices in ascending order
// (== np.where(...)[0]) — ordered stream compaction by wave ballots + a 4-entry LDS prefix.
// ranges != NULL: grasp g only sees points [ranges[2g], ranges[2g] + ranges[2g+1]) of a cloud ARENA (many scene /
// object clouds resident in one buffer).  gather != NULL: grasp g sees the Pg points arena[gather[g][0..Pg)] (a
// per-sample random subsample of stacked views, dataset.py:252-254).  Either way the indices written are
// arena-absolute and in the order of the grasp's own cloud, so crop_resample runs unchanged.
// NT threads per workgroup: 256 when there are many grasps (config 5: 100,000 workgroups), 1024 when there are few (a
// training batch of 64: the scan of a 20,000-point view is 5 trips instead of 20, each one memory latency).
template <bool F64, int NT>
__global__ __launch_bounds__(NT) void crop_count_compact_kernel(
    const void *__restrict__ cloud, int P, const double *__restrict__ frames, const int *__restrict__ ranges,
    const int *__restrict__ gather, int Pg, int max_keep, int *__restrict__ counts, int *__restrict__ idx,
    const int *__restrict__ item, int *__restrict__ seg_counts = nullptr, int phase = 0) {
    // Segmented form (gridDim.y = SEG > 1, a training batch of few samples with LONG clouds — 64 x 50,000 gathered rows in
    // the full-view datasets): a sample's rows are cut into SEG consecutive segments, one workgroup each, in two launches —
    // phase 1 counts a segment's in-box rows into seg_counts (g, SEG); phase 2 scans again and writes from the offset the
    // preceding segments' counts give, so the list keeps the single-pass order.  Twice the loads, SEG x the workgroups:
    // the scan is a chain of dependent random reads, i.e. latency (0.12 -> ~0.04 ms per full-view batch).
    // UNR blocks of 256 points per trip: the UNR loads of a thread are independent, so a trip costs ONE memory latency
    // and one barrier pair instead of UNR of each (a training batch is 64 workgroups on 256 CUs — latency-, not
    // throughput-bound); positions stay in ascending point order (block-major, then wave, then lane)
    constexpr int UNR = 4;
    __shared__ int wcnt[UNR][NT / 64];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Frame F;
    load_frame(frames + (size_t)(item ? item[g] : g) * 18, F);   // item: frames is a per-dataset table, g's row = item[g]
    int running = 0;
    int *out = idx + (size_t)g * max_keep;
    const int p_begin = (!gather && ranges) ? ranges[2 * g] : 0;
    const int n_all = gather ? Pg : (ranges ? ranges[2 * g + 1] : P);
    const int *gi = gather ? gather + (size_t)g * Pg : nullptr;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int SEG = gridDim.y, seg = blockIdx.y;
    const int seg_len = (n_all + SEG - 1) / SEG;
    const int i0 = seg * seg_len;
    const int n = i0 + seg_len < n_all ? i0 + seg_len : n_all;      // this workgroup scans rows [i0, n)
    if (phase == 2)
        for (int s2 = 0; s2 < seg; ++s2) running += seg_counts[(size_t)g * SEG + s2];
    if (phase == 1) {                                               // count only: no order needed, no barriers per trip
        int c = 0;
#pragma unroll 4
        for (int i = i0 + tid; i < n; i += NT) {
            const int p = gi ? gi[i] : p_begin + i;
            double x, y, z, a, b, cc;
            load_point<F64>(cloud, p, x, y, z);
            to_frame(F, x, y, z, a, b, cc);
            c += ((a > F.lo[0]) && (a < F.hi[0]) && (b > F.lo[1]) && (b < F.hi[1]) && (cc > F.lo[2]) && (cc < F.hi[2])) ? 1 : 0;
        }
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k);
        if (lane == 0) wcnt[0][wave] = c;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < NT / 64; ++w) t += wcnt[0][w];
            seg_counts[(size_t)g * SEG + seg] = t;
        }
        return;
    }
    for (int base = i0; base < n; base += NT * UNR) {
        bool in[UNR];
        int p[UNR];
        unsigned long long mask[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int i = base + j * NT + tid;
            p[j] = i < n ? (gi ? gi[i] : p_begin + i) : 0;
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const int i = base + j * NT + tid;
            in[j] = false;
            if (i < n) {
                double x, y, z, a, b, c;
                load_point<F64>(cloud, p[j], x, y, z);
                to_frame(F, x, y, z, a, b, c);
                in[j] = (a > F.lo[0]) && (a < F.hi[0]) && (b > F.lo[1]) && (b < F.hi[1]) && (c > F.lo[2]) && (c < F.hi[2]);
            }
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            mask[j] = __ballot(in[j]);
            if (lane == 0) wcnt[j][wave] = __popcll(mask[j]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            int woff = 0, total = 0;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) { const int c = wcnt[j][w]; if (w < wave) woff += c; total += c; }
            if (in[j]) {
                const int pos = running + woff + __popcll(mask[j] & below);
                if (pos < max_keep) out[pos] = p[j];
            }
            running += total;
        }
        __syncthreads();
    }
    if (phase == 2) {
        if (tid == 0 && seg == 0) {
            int t = 0;
            for (int s2 = 0; s2 < SEG; ++s2) t += seg_counts[(size_t)g * SEG + s2];
            counts[g] = t;
        }
        return;
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

// Block-wide sum over 256 threads (every thread gets the total).
__device__ __forceinline__ int crop_block_sum(int v, int *sh) {
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) v += __shfl_xor(v, k);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// One workgroup per grasp: pick N of the m = min(count, max_keep) kept points and write them in the
// hand frame as (3,N) fp32.  Rule of the reference: WITHOUT replacement iff m > N (mode 0, dataset.py:439)
// or m >= N (mode 1, kinect2grasp.py:474); otherwise WITH replacement.  sel != NULL injects the draw:
// sel[g][n] in [0, m) is the rank of the kept point to take (tests / bit-reproducible pipelines).
//
// Without replacement = a uniform random N-subset: every kept point gets an iid 32-bit key (counter hash of
// (seed, grasp, rank)); the N smallest keys win, exact ties at the cut resolved towards the lower rank.  The N-th
// smallest key is found by bisection over the key space with block-wide counts (32 rounds over keys parked in
// LDS), and the winners leave in ascending rank by ballot-ordered compaction — fully parallel, where a
// Fisher-Yates draw is a serial chain of N dependent swaps.  The column ORDER of the output is therefore the
// index order, not a random permutation; the scorer is invariant to it (per-point MLP + max-pool).
//
// count > max_keep (the index list was truncated): the draw must still be uniform over ALL in-box points
// (kinect2grasp.py:473-478, dataset.py:438-444), so the kernel re-scans the grasp's own cloud instead of using the
// list.  Without replacement: the same keys (counter hash of the in-box RANK) are recomputed on the fly and the
// N-th smallest is found by a 4-level radix histogram (4 scans) + 1 emitting scan.  With replacement (only
// possible when max_keep < count <= N): the full list (count <= N entries) is rebuilt in LDS first.
// The body of the resample step for grasp g = blockIdx.x (one workgroup): ``cnt`` in-box points, the first
// min(cnt, max_keep) of them listed in ``gi`` (global memory — crop_resample_kernel — or the workgroup's own LDS —
// crop_indexed_kernel, where the list never leaves the CU); ``keys``: max(max_keep, N) words of LDS scratch.
template <bool F64>
__device__ __forceinline__ void crop_resample_body(
    const void *__restrict__ cloud, int P, const double *__restrict__ frames, const int *__restrict__ ranges,
    const int *__restrict__ gather, int Pg, const int cnt, const int *gi, int max_keep, int N, int mode, int min_points,
    unsigned long long seed, long long g_base, const int *__restrict__ rows, const int *__restrict__ sel,
    float *__restrict__ out, unsigned char *__restrict__ valid, const int *__restrict__ item, unsigned *keys) {
    __shared__ int shi[4];
    __shared__ int wsel[4], wtie[4];
    __shared__ int hist[256];
    __shared__ int hsel[2];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int m = cnt < max_keep ? cnt : max_keep;
    // the draw of a grasp is keyed by its GLOBAL index: a candidate scores the same whichever shard / scoring batch
    // it lands in (kinect2grasp.py:454-497 scores every candidate independently of the others)
    const unsigned long long gkey = (unsigned long long)(g_base + (long long)g) << 32;
    const int row = rows ? rows[g] : g;       // rows: destination row of the compacted batch, < 0 = dropped
    const bool ok = cnt >= min_points && m > 0;
    if (tid == 0) valid[g] = ok ? 1 : 0;
    if (row < 0) return;
    float *o = out + (size_t)row * 3 * N;
    if (!ok) {
        for (int i = tid; i < 3 * N; i += 256) o[i] = 0.f;
        return;
    }
    Frame F;
    load_frame(frames + (size_t)(item ? item[g] : g) * 18, F);
    if (cnt > max_keep && !sel) {
        // ---- overflow: scan the grasp's own cloud (same order and test as crop_count_compact_kernel)
        const int p_begin = (!gather && ranges) ? ranges[2 * g] : 0;
        const int n = gather ? Pg : (ranges ? ranges[2 * g + 1] : P);
        const int *gl = gather ? gather + (size_t)g * Pg : nullptr;
        auto key_of = [&](int rank) {
            return (unsigned)(mix64(seed ^ mix64(gkey | (unsigned)rank)) >> 32);
        };
        // scan(f): f(in, rank, p) for every candidate point, rank = number of in-box points before it
        auto scan = [&](auto f) {
            int running = 0;
            for (int base = 0; base < n; base += 256) {
                const int i = base + tid;
                bool in = false; int p = 0;
                if (i < n) {
                    p = gl ? gl[i] : p_begin + i;
                    double x, y, z, a, b, c;
                    load_point<F64>(cloud, p, x, y, z);
                    to_frame(F, x, y, z, a, b, c);
                    in = (a > F.lo[0]) && (a < F.hi[0]) && (b > F.lo[1]) && (b < F.hi[1]) && (c > F.lo[2]) && (c < F.hi[2]);
                }
                const unsigned long long mask = __ballot(in);
                __syncthreads();
                if (lane == 0) wsel[wave] = __popcll(mask);
                __syncthreads();
                int woff = 0, total = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) { const int c = wsel[w]; if (w < wave) woff += c; total += c; }
                f(in, running + woff + __popcll(mask & ((1ull << lane) - 1ull)), p);
                running += total;
            }
        };
        const bool without_all = (mode == 0) ? (cnt > N) : (cnt >= N);
        if (!without_all) {   // with replacement and max_keep < cnt <= N: rebuild the complete list in LDS
            int *list = (int *)keys;
            scan([&](bool in, int rank, int p) { if (in) list[rank] = p; });
            __syncthreads();
            for (int nn = tid; nn < N; nn += 256) {
                const int r = (int)(mix64(seed ^ mix64(gkey | (unsigned)nn)) % (unsigned long long)cnt);
                double x, y, z, a, b, c;
                load_point<F64>(cloud, list[r], x, y, z);
                to_frame(F, x, y, z, a, b, c);
                o[nn] = (float)a; o[N + nn] = (float)b; o[2 * N + nn] = (float)c;
            }
            return;
        }
        // ---- one geometric scan (round 6).  Dense scenes put EVERY hand here: the box of the Robotiq hand holds 22 k ...
        // 49.7 k of the 50 k points of a table-top object (BASELINE configs[4] with sampled candidates), and the radix
        // path below tests every point of the cloud five times in fp64 (1.02 ms per 1,024 hands).  Here: (A) one scan
        // leaves the in-box bitmask of the grasp's cloud in LDS; (B) a walk over the set bits hashes each in-box rank
        // and keeps the keys under a cut chosen so that N + 8 sqrt(N) + 64 of them are expected (fewer than N: 1e-19)
        // as (key, position) records; the N-th smallest key T follows by bisection over those ~1,350 records instead of
        // 4 histogram scans of the cloud; (C) the winners' bits are set, a popcount scan gives every winner its column:
        // ascending position = ascending rank, ties at T to the lowest ranks — the radix path's output, bit for bit.
        // Falls through to the radix path when the scratch cannot hold the bitmask + records, or the cut missed.
        {
            const int KW = max_keep > N ? max_keep : N;                        // words of `keys`
            const int nwords = (n + 63) >> 6;
            const double E = (double)N + 8.0 * sqrt((double)N) + 64.0;
            const int capw = (int)(E * 0.25 + 10.0 * sqrt(E * 0.25) + 16.0);   // records per wave (+10 sigma)
            bool fast = (long long)3 * nwords + (long long)8 * capw <= (long long)KW;
            if (fast) {
                unsigned *maskw = keys;                                        // [nwords][2] bit i of the scan order
                int *pre = (int *)(keys + 2 * nwords);                         // [nwords] exclusive popcount prefix
                unsigned *cand = keys + 3 * nwords;                            // [4 waves][capw][2] = key, position
                const unsigned long long lt = (1ull << lane) - 1ull;
                auto word = [&](int wi) { return (unsigned long long)maskw[2 * wi] | ((unsigned long long)maskw[2 * wi + 1] << 32); };
                auto prefix_scan = [&]() {                                      // pre[wi] = set bits before word wi
                    const int W = (nwords + 255) >> 8;
                    int loc = 0;
                    for (int q = 0; q < W; ++q) { const int wi = tid * W + q; if (wi < nwords) loc += __popcll(word(wi)); }
                    int incl = loc;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
                    __syncthreads();
                    if (lane == 63) shi[wave] = incl;
                    __syncthreads();
                    int run = incl - loc;
#pragma unroll
                    for (int w = 0; w < 4; ++w) if (w < wave) run += shi[w];
                    for (int q = 0; q < W; ++q) { const int wi = tid * W + q; if (wi < nwords) { pre[wi] = run; run += __popcll(word(wi)); } }
                    __syncthreads();
                };
                // (A) the only pass that touches the cloud: wave w owns words w, w + 4, ...  (four words per trip with their
                // points requested together: measured, no gain — the workgroup's 0.2 ms is spread over all four phases)
                for (int wi = wave; wi < nwords; wi += 4) {
                    const int i = wi * 64 + lane;
                    bool in = false;
                    if (i < n) {
                        const int p = gl ? gl[i] : p_begin + i;
                        double x, y, z, a, b, c;
                        load_point<F64>(cloud, p, x, y, z);
                        to_frame(F, x, y, z, a, b, c);
                        in = (a > F.lo[0]) && (a < F.hi[0]) && (b > F.lo[1]) && (b < F.hi[1]) && (c > F.lo[2]) && (c < F.hi[2]);
                    }
                    const unsigned long long mk = __ballot(in);
                    if (lane == 0) { maskw[2 * wi] = (unsigned)mk; maskw[2 * wi + 1] = (unsigned)(mk >> 32); }
                }
                __syncthreads();
                prefix_scan();
                // (B) keys of the in-box ranks; records under the cut, per wave, ballot-compacted
                const double cut = E / (double)cnt * 4294967296.0;
                const unsigned T_hi = cut >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)cut;
                unsigned *cw = cand + (size_t)wave * capw * 2;
                int wc = 0;
                bool over = false;
                for (int wi = wave; wi < nwords; wi += 4) {
                    const unsigned long long mk = word(wi);
                    const bool in = (mk >> lane) & 1ull;
                    unsigned k = 0xFFFFFFFFu;
                    if (in) k = key_of(pre[wi] + __popcll(mk & lt));
                    const bool c = in && k <= T_hi;
                    const unsigned long long cm = __ballot(c);
                    if (cm) {
                        const int nadd = __popcll(cm);
                        if (wc + nadd > capw) { over = true; break; }
                        if (c) { const int sl = wc + __popcll(cm & lt); cw[2 * sl] = k; cw[2 * sl + 1] = (unsigned)(wi * 64 + lane); }
                        wc += nadd;
                    }
                }
                if (lane == 0) wsel[wave] = over ? -1 : wc;
                __syncthreads();
                int C = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) { if (wsel[w] < 0) fast = false; else C += wsel[w]; }
                if (C < N) fast = false;
                if (fast) {
                    // pred(key, pos) counted over all records, block-wide
                    auto count_if = [&](auto pred) {
                        int c = 0;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const unsigned *r = cand + (size_t)w * capw * 2;
                            const int nw = wsel[w];
                            for (int i = tid; i < nw; i += 256) c += pred(r[2 * i], r[2 * i + 1]) ? 1 : 0;
                        }
                        return crop_block_sum(c, shi);
                    };
                    unsigned lo = 0u, hi = T_hi;                               // smallest T with #(key <= T) >= N
                    while (lo < hi) {
                        const unsigned mid = lo + ((hi - lo) >> 1);
                        if (count_if([&](unsigned k, unsigned) { return k <= mid; }) >= N) hi = mid; else lo = mid + 1u;
                    }
                    const unsigned T = lo;
                    const int below = count_if([&](unsigned k, unsigned) { return k < T; });
                    const int need = N - below;                                // of the key == T records: the lowest positions
                    unsigned R = 0xFFFFFFFFu;
                    if (count_if([&](unsigned k, unsigned) { return k == T; }) > need) {
                        unsigned l2 = 0u, h2 = (unsigned)n;                    // smallest R with #(key == T, pos <= R) >= need
                        while (l2 < h2) {
                            const unsigned mid = l2 + ((h2 - l2) >> 1);
                            if (count_if([&](unsigned k, unsigned q) { return k == T && q <= mid; }) >= need) h2 = mid; else l2 = mid + 1u;
                        }
                        R = l2;
                    }
                    // (C) winners' bits, then every winner's column
                    for (int wi = tid; wi < 2 * nwords; wi += 256) maskw[wi] = 0u;
                    __syncthreads();
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const unsigned *r = cand + (size_t)w * capw * 2;
                        const int nw = wsel[w];
                        for (int i = tid; i < nw; i += 256) {
                            const unsigned k = r[2 * i], q = r[2 * i + 1];
                            if (k < T || (k == T && q <= R)) atomicOr(&maskw[q >> 5], 1u << (q & 31u));
                        }
                    }
                    __syncthreads();
                    prefix_scan();
                    for (int wi = wave; wi < nwords; wi += 4) {
                        const unsigned long long mk = word(wi);
                        if ((mk >> lane) & 1ull) {
                            const int nn = pre[wi] + __popcll(mk & lt);
                            const int i = wi * 64 + lane;
                            const int p = gl ? gl[i] : p_begin + i;
                            double x, y, z, a, b, c;
                            load_point<F64>(cloud, p, x, y, z);
                            to_frame(F, x, y, z, a, b, c);
                            o[nn] = (float)a; o[N + nn] = (float)b; o[2 * N + nn] = (float)c;
                        }
                    }
                    return;
                }
                __syncthreads();      // the radix path reuses the scratch
            }
        }
        // N-th smallest key by radix selection: prefix = the key bits fixed so far, need = how many keys with
        // that prefix are still to be taken
        unsigned prefix = 0u; int need = N;
        for (int level = 0; level < 4; ++level) {
            const int shift = 24 - 8 * level;
            hist[tid] = 0;
            __syncthreads();
            scan([&](bool in, int rank, int) {
                if (in) {
                    const unsigned k = key_of(rank);
                    if (level == 0 || (k >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(k >> shift) & 255u], 1);
                }
            });
            __syncthreads();
            if (tid == 0) {   // 256 bins: a serial walk is cheaper than a scan
                int acc = 0, b = 0;
                for (; b < 255 && acc + hist[b] < need; ++b) acc += hist[b];
                hsel[0] = b; hsel[1] = need - acc;
            }
            __syncthreads();
            prefix |= (unsigned)hsel[0] << shift;
            need = hsel[1];
            __syncthreads();
        }
        const unsigned T = prefix;   // keys < T are all taken; of the keys == T the `need` lowest ranks
        int run_sel = 0, run_tie = 0;
        scan([&](bool in, int rank, int p) {
            const unsigned k = in ? key_of(rank) : 0xFFFFFFFFu;
            const bool tie = in && k == T;
            const unsigned long long tmask = __ballot(tie);
            __syncthreads();
            if (lane == 0) wtie[wave] = __popcll(tmask);
            __syncthreads();
            int toff = 0, ttot = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const int c = wtie[w]; if (w < wave) toff += c; ttot += c; }
            const int tie_rank = run_tie + toff + __popcll(tmask & ((1ull << lane) - 1ull));
            const bool take = in && (k < T || (tie && tie_rank < need));
            const unsigned long long smask = __ballot(take);
            __syncthreads();
            if (lane == 0) wtie[wave] = __popcll(smask);
            __syncthreads();
            int soff = 0, stot = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const int c = wtie[w]; if (w < wave) soff += c; stot += c; }
            if (take) {
                const int nn = run_sel + soff + __popcll(smask & ((1ull << lane) - 1ull));
                double x, y, z, a, b, c;
                load_point<F64>(cloud, p, x, y, z);
                to_frame(F, x, y, z, a, b, c);
                o[nn] = (float)a; o[N + nn] = (float)b; o[2 * N + nn] = (float)c;
            }
            run_sel += stot; run_tie += ttot;
        });
        return;
    }
    const bool without = (mode == 0) ? (m > N) : (m >= N);
    if (!sel && without) {
        for (int i = tid; i < m; i += 256)
            keys[i] = (unsigned)(mix64(seed ^ mix64(gkey | (unsigned)i)) >> 32);
        __syncthreads();
        auto count_le = [&](unsigned T) {
            int c = 0;
            for (int i = tid; i < m; i += 256) c += keys[i] <= T ? 1 : 0;
            return crop_block_sum(c, shi);
        };
        // (a 4-level radix select over these LDS keys — histogram + bin walk per byte, as the overflow path above — was
        // measured here and is SLOWER at config-5 scale: 3,248 keys per hand contend on 256 LDS bins, resample of
        // 100,000 hands 5.5 -> 6.2 ms; and a training batch does not notice either way)
        unsigned lo = 0u, hi = 0xFFFFFFFFu;            // smallest T with #(key <= T) >= N   (m >= N here)
        while (lo < hi) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            if (count_le(mid) >= N) hi = mid; else lo = mid + 1u;
        }
        const unsigned T = lo;
        const int need = N - (T ? count_le(T - 1u) : 0);   // how many of the key == T ties are taken (lowest ranks)
        int run_sel = 0, run_tie = 0;
        for (int base = 0; base < m; base += 256) {
            const int i = base + tid;
            const unsigned k = i < m ? keys[i] : 0xFFFFFFFFu;
            const bool tie = i < m && k == T;
            const unsigned long long tmask = __ballot(tie);
            if (lane == 0) wtie[wave] = __popcll(tmask);
            __syncthreads();
            int toff = 0, ttot = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const int c = wtie[w]; if (w < wave) toff += c; ttot += c; }
            const int tie_rank = run_tie + toff + __popcll(tmask & ((1ull << lane) - 1ull));
            const bool take = i < m && (k < T || (tie && tie_rank < need));
            const unsigned long long smask = __ballot(take);
            if (lane == 0) wsel[wave] = __popcll(smask);
            __syncthreads();
            int soff = 0, stot = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const int c = wsel[w]; if (w < wave) soff += c; stot += c; }
            if (take) {
                const int n = run_sel + soff + __popcll(smask & ((1ull << lane) - 1ull));
                double x, y, z, a, b, c;
                load_point<F64>(cloud, gi[i], x, y, z);
                to_frame(F, x, y, z, a, b, c);
                o[n] = (float)a; o[N + n] = (float)b; o[2 * N + n] = (float)c;
            }
            run_sel += stot; run_tie += ttot;
            __syncthreads();
        }
        return;
    }
    for (int n = tid; n < N; n += 256) {
        int r;
        if (sel) {
            r = sel[(size_t)g * N + n];
            r = r < 0 ? 0 : (r >= m ? m - 1 : r);
        } else {
            r = (int)(mix64(seed ^ mix64(gkey | (unsigned)n)) % (unsigned long long)m);
        }
        double x, y, z, a, b, c;
        load_point<F64>(cloud, gi[r], x, y, z);
        to_frame(F, x, y, z, a, b, c);
        o[n] = (float)a; o[N + n] = (float)b; o[2 * N + n] = (float)c;
    }
}

template <bool F64>
__global__ __launch_bounds__(256) void crop_resample_kernel(
    const void *__restrict__ cloud, int P, const double *__restrict__ frames, const int *__restrict__ ranges,
    const int *__restrict__ gather, int Pg, const int *__restrict__ counts,
    const int *__restrict__ idx, int max_keep, int N, int mode, int min_points, unsigned long long seed,
    long long g_base, const int *__restrict__ rows, const int *__restrict__ sel, float *__restrict__ out,
    unsigned char *__restrict__ valid, const int *__restrict__ item) {
    extern __shared__ unsigned keys[];   // [max(max_keep, N)]: keys of the without-replacement draw / rebuilt list
    crop_resample_body<F64>(cloud, P, frames, ranges, gather, Pg, counts[blockIdx.x], idx + (size_t)blockIdx.x * max_keep,
                            max_keep, N, mode, min_points, seed, g_base, rows, sel, out, valid, item, keys);
}

// ---------------------------------------------------------------------------------------------------------------
// Crop over a spatial index (gpg.CloudIndex: the cloud re-ordered along a Morton curve + the bounding sphere of every
// 64-point chunk) — BASELINE configs[4], where 100,000 candidate hands are cropped out of ONE 50,000-point scene
// (kinect2grasp.py:238-258).  A hand's box holds a few percent of the scene: lane c tests chunk c's sphere against the
// box in the hand frame, only the surviving chunks' points are evaluated (the same fp64 per-point test), in ascending
// SORTED position — so a list is the in-box points in the order of the sorted cloud, which is also the order a plain scan
// of that cloud visits them (the overflow path of the resample step stays consistent).
//   crop_count_compact_indexed_kernel  counts + lists in global memory (the API twin of crop_count_compact_kernel);
//   crop_indexed_kernel                count -> list in the workgroup's LDS -> resample, ONE launch per batch of grasps:
//                                      the index lists (4.8 k entries per hand on the bench scene, ~1 GB per launch
//                                      written and read back) never leave the CU.
template <bool F64>
__device__ __forceinline__ int crop_indexed_list(const void *__restrict__ cloud, int P, const double *__restrict__ spheres,
                                                 int C, const Frame &F, int max_keep, int *list /* LDS or global */) {
    __shared__ int ch_list[256];
    __shared__ int ch_cnt[4];
    __shared__ int pt_cnt[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int running = 0;
    // A world-space sphere of radius r covers [a - r |m_i|, a + r |m_i|] along frame axis i: scale by the row norms, so
    // the cull stays conservative for ANY (G,18) frame this public entry is handed (a mesh -> cloud transform with
    // scale, un-normalised axes), not only for the unit rows crop.py builds.
    const double rn0 = sqrt(F.m[0] * F.m[0] + F.m[1] * F.m[1] + F.m[2] * F.m[2]) * (1.0 + 1e-9);
    const double rn1 = sqrt(F.m[3] * F.m[3] + F.m[4] * F.m[4] + F.m[5] * F.m[5]) * (1.0 + 1e-9);
    const double rn2 = sqrt(F.m[6] * F.m[6] + F.m[7] * F.m[7] + F.m[8] * F.m[8]) * (1.0 + 1e-9);
    for (int cbase = 0; cbase < C; cbase += 256) {
        // ---- broad phase: 256 chunks, ordered compaction of the survivors into ch_list
        const int c = cbase + tid;
        bool pass = false;
        if (c < C) {
            const double4 sp = *(const double4 *)(spheres + (size_t)c * 4);
            double a, b, cc;
            to_frame(F, sp.x, sp.y, sp.z, a, b, cc);
            const double r = sp.w * (1.0 + 1e-9) + 1e-12;
            const double r0 = r * rn0, r1 = r * rn1, r2 = r * rn2;
            pass = a + r0 > F.lo[0] && a - r0 < F.hi[0] && b + r1 > F.lo[1] && b - r1 < F.hi[1] && cc + r2 > F.lo[2] &&
                   cc - r2 < F.hi[2];
        }
        const unsigned long long pm = __ballot(pass);
        __syncthreads();                                    // (the previous round's readers of ch_list are done)
        if (lane == 0) ch_cnt[wave] = __popcll(pm);
        __syncthreads();
        int woff = 0, nch = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int n = ch_cnt[w]; if (w < wave) woff += n; nch += n; }
        if (pass) ch_list[woff + __popcll(pm & below)] = c;
        __syncthreads();
        // ---- narrow phase: 16 chunks a trip (wave w takes chunks 4 j + w of the trip, j = 0..3), exact per-point test
        for (int t0 = 0; t0 < nch; t0 += 16) {
            bool in[4]; int p[4]; unsigned long long mask[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = t0 + j * 4 + wave;
                in[j] = false; p[j] = 0;
                if (e < nch) {
                    p[j] = ch_list[e] * 64 + lane;
                    if (p[j] < P) {
                        double x, y, z, a, b, cc;
                        load_point<F64>(cloud, p[j], x, y, z);
                        to_frame(F, x, y, z, a, b, cc);
                        in[j] = (a > F.lo[0]) && (a < F.hi[0]) && (b > F.lo[1]) && (b < F.hi[1]) && (cc > F.lo[2]) && (cc < F.hi[2]);
                    }
                }
                mask[j] = __ballot(in[j]);
                if (lane == 0) pt_cnt[j][wave] = __popcll(mask[j]);
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int off = 0, tot = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) { const int n = pt_cnt[j][w]; if (w < wave) off += n; tot += n; }
                if (in[j]) {
                    const int pos = running + off + __popcll(mask[j] & below);
                    if (pos < max_keep) list[pos] = p[j];
                }
                running += tot;
            }
            __syncthreads();
        }
    }
    return running;
}

template <bool F64>
__global__ __launch_bounds__(256) void crop_count_compact_indexed_kernel(
    const void *__restrict__ cloud, int P, const double *__restrict__ spheres, int C, const double *__restrict__ frames,
    int max_keep, int *__restrict__ counts, int *__restrict__ idx) {
    Frame F;
    load_frame(frames + (size_t)blockIdx.x * 18, F);
    const int n = crop_indexed_list<F64>(cloud, P, spheres, C, F, max_keep, idx + (size_t)blockIdx.x * max_keep);
    if (threadIdx.x == 0) counts[blockIdx.x] = n;
}

template <bool F64>
__global__ __launch_bounds__(256) void crop_indexed_kernel(
    const void *__restrict__ cloud, int P, const double *__restrict__ spheres, int C, const double *__restrict__ frames,
    int max_keep, int N, int mode, int min_points, unsigned long long seed, long long g_base, int *__restrict__ counts,
    float *__restrict__ out, unsigned char *__restrict__ valid) {
    extern __shared__ unsigned dyn[];                       // keys[max(max_keep, N)] | list[max_keep]
    unsigned *keys = dyn;
    int *list = (int *)(dyn + (max_keep > N ? max_keep : N));
    Frame F;
    load_frame(frames + (size_t)blockIdx.x * 18, F);
    const int n = crop_indexed_list<F64>(cloud, P, spheres, C, F, max_keep, list);
    if (threadIdx.x == 0) counts[blockIdx.x] = n;
    __syncthreads();                                        // the list is complete
    crop_resample_body<F64>(cloud, P, frames, nullptr, nullptr, 0, n, list, max_keep, N, mode, min_points, seed, g_base,
                            nullptr, nullptr, out, valid, nullptr, keys);
}

// my_collate (main_1v.py:48-50) on the device: sample g of a training batch is kept iff its crop holds at least
// min_points points (dataset.py:71-72) and it carries a label (dataset.py:446-453: None between the thresholds).
// One workgroup: an ordered prefix over keep[] gives each kept sample its row in the compacted batch (rows[g], -1 =
// dropped; pngpd_crop_resample then writes straight to that row), compacts the labels, and leaves the kept count in
// device memory — the host learns it from a pinned copy one batch later, never by stalling on this one.
constexpr int KEEP_NT = PNGPD_ASAN ? 256 : 1024;      // (sanitizer builds: see pngpd_common.h on 1024-thread workgroups)
__global__ __launch_bounds__(KEEP_NT) void batch_keep_rows_kernel(const int *__restrict__ counts,
                                                               const long long *__restrict__ labels, int G,
                                                               int min_points, int *__restrict__ rows,
                                                               long long *__restrict__ labels_out,
                                                               int *__restrict__ n_keep,
                                                               const int *__restrict__ item) {
    __shared__ int wcnt[KEEP_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int running = 0;
    for (int base = 0; base < G; base += KEEP_NT) {
        const int g = base + tid;
        const long long lab = g < G ? labels[item ? item[g] : g] : -1;
        const bool keep = g < G && counts[g] >= min_points && counts[g] > 0 && lab >= 0;
        const unsigned long long mask = __ballot(keep);
        __syncthreads();
        if (lane == 0) wcnt[wave] = __popcll(mask);
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < KEEP_NT / 64; ++w) { const int c = wcnt[w]; if (w < wave) woff += c; total += c; }
        if (g < G) {
            const int r = running + woff + __popcll(mask & ((1ull << lane) - 1ull));
            rows[g] = keep ? r : -1;
            if (keep) labels_out[r] = lab;
        }
        running += total;
    }
    if (tid == 0) *n_keep = running;
}

// The per-sample cloud of the full-view datasets (dataset.py:252-254): ``pc[np.random.choice(len(pc), size=Pg)]`` of
// the stack of the k view files drawn for the sample == Pg uniform rows of the stacked length L, each mapped to the
// arena through the sample's (start, len) spans.  One thread per drawn row, a counter hash of (seed, global sample
// index, draw) as the generator; (hash * L) >> 64 is the unbiased-to-2^-64 uniform integer in [0, L).
__global__ __launch_bounds__(256) void stack_gather_lists_kernel(const int *__restrict__ spans, int k, int Pg,
                                                                 unsigned long long seed, long long g_base,
                                                                 int *__restrict__ gather) {
    __shared__ long long cum[65];
    __shared__ int start[64];
    const int g = blockIdx.y, tid = threadIdx.x;
    if (tid == 0) {
        long long acc = 0;
        for (int v = 0; v < k; ++v) {
            cum[v] = acc;
            start[v] = spans[((size_t)g * k + v) * 2];
            acc += spans[((size_t)g * k + v) * 2 + 1];
        }
        cum[k] = acc;
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + tid;
    if (j >= Pg) return;
    const unsigned long long L = (unsigned long long)cum[k];
    const unsigned long long h = mix64(seed ^ mix64(((unsigned long long)(g_base + g) << 32) | (unsigned)j));
    const long long r = (long long)__umul64hi(h, L);
    int v = 0;
    while (v + 1 < k && cum[v + 1] <= r) ++v;
    gather[(size_t)g * Pg + j] = start[v] + (int)(r - cum[v]);
}

extern "C" {

int pngpd_crop_count_compact(const void *cloud, int cloud_is_f64, int P, const double *frames, int G,
                             int max_keep, int *counts, int *idx, void *stream) {
    if (!cloud || !frames || !counts || !idx || P <= 0 || G <= 0 || max_keep <= 0) return PNGPD_ERR_INVALID_ARG;
    if (cloud_is_f64)
        hipLaunchKernelGGL((crop_count_compact_kernel<true, 256>), dim3(G), dim3(256), 0, (hipStream_t)stream,
                           cloud, P, frames, (const int *)nullptr, (const int *)nullptr, 0, max_keep, counts, idx, (const int *)nullptr);
    else
        hipLaunchKernelGGL((crop_count_compact_kernel<false, 256>), dim3(G), dim3(256), 0, (hipStream_t)stream,
                           cloud, P, frames, (const int *)nullptr, (const int *)nullptr, 0, max_keep, counts, idx, (const int *)nullptr);
    return pngpd_launch_status();
}

int pngpd_crop_count_compact_ranges(const void *arena, int cloud_is_f64, int P, const double *frames,
                                    const int *ranges, int G, int max_keep, int *counts, int *idx, void *stream) {
    if (!arena || !frames || !ranges || !counts || !idx || P <= 0 || G <= 0 || max_keep <= 0)
        return PNGPD_ERR_INVALID_ARG;
    if (cloud_is_f64)
        hipLaunchKernelGGL((crop_count_compact_kernel<true, 256>), dim3(G), dim3(256), 0, (hipStream_t)stream,
                           arena, P, frames, ranges, (const int *)nullptr, 0, max_keep, counts, idx, (const int *)nullptr);
    else
        hipLaunchKernelGGL((crop_count_compact_kernel<false, 256>), dim3(G), dim3(256), 0, (hipStream_t)stream,
                           arena, P, frames, ranges, (const int *)nullptr, 0, max_keep, counts, idx, (const int *)nullptr);
    return pngpd_launch_status();
}

int pngpd_crop_count_compact_gather(const void *arena, int cloud_is_f64, int P, const double *frames,
                                    const int *gather, int Pg, int G, int max_keep, int *counts, int *idx,
                                    void *stream) {
    if (!arena || !frames || !gather || !counts || !idx || P <= 0 || Pg <= 0 || G <= 0 || max_keep <= 0)
        return PNGPD_ERR_INVALID_ARG;
    if (cloud_is_f64)
        hipLaunchKernelGGL((crop_count_compact_kernel<true, 256>), dim3(G), dim3(256), 0, (hipStream_t)stream,
                           arena, P, frames, (const int *)nullptr, gather, Pg, max_keep, counts, idx, (const int *)nullptr);
    else
        hipLaunchKernelGGL((crop_count_compact_kernel<false, 256>), dim3(G), dim3(256), 0, (hipStream_t)stream,
                           arena, P, frames, (const int *)nullptr, gather, Pg, max_keep, counts, idx, (const int *)nullptr);
    return pngpd_launch_status();
}

int pngpd_crop_resample(const void *cloud, int cloud_is_f64, int P, const double *frames, const int *ranges,
                        const int *gather, int Pg, int G, const int *counts, const int *idx, int max_keep, int N,
                        int mode, int min_points, unsigned long long seed, long long g_base, const int *rows,
                        const int *sel, float *out, unsigned char *valid, void *stream) {
    if (!cloud || !frames || !counts || !idx || !out || !valid || P <= 0 || G <= 0 || max_keep <= 0 || N <= 0 ||
        (mode != 0 && mode != 1) || (gather && Pg <= 0))
        return PNGPD_ERR_INVALID_ARG;
    const size_t lds = (size_t)(max_keep > N ? max_keep : N) * sizeof(int);
    if (lds > 150 * 1024) return PNGPD_ERR_UNSUPPORTED;
    if (cloud_is_f64) {
        const int st = pngpd_allow_lds((const void *)crop_resample_kernel<true>, lds);
        if (st != PNGPD_OK) return st;
        hipLaunchKernelGGL(crop_resample_kernel<true>, dim3(G), dim3(256), lds, (hipStream_t)stream,
                           cloud, P, frames, ranges, gather, Pg, counts, idx, max_keep, N, mode, min_points, seed, g_base,
                           rows, sel, out, valid, (const int *)nullptr);
    } else {
        const int st = pngpd_allow_lds((const void *)crop_resample_kernel<false>, lds);
        if (st != PNGPD_OK) return st;
        hipLaunchKernelGGL(crop_resample_kernel<false>, dim3(G), dim3(256), lds, (hipStream_t)stream,
                           cloud, P, frames, ranges, gather, Pg, counts, idx, max_keep, N, mode, min_points, seed, g_base,
                           rows, sel, out, valid, (const int *)nullptr);
    }
    return pngpd_launch_status();
}

int pngpd_batch_keep_rows(const int *counts, const long long *labels, int G, int min_points, int *rows,
                          long long *labels_out, int *n_keep, void *stream) {
    if (!counts || !labels || !rows || !labels_out || !n_keep || G <= 0) return PNGPD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(batch_keep_rows_kernel, dim3(1), dim3(KEEP_NT), 0, (hipStream_t)stream, counts, labels, G,
                       min_points, rows, labels_out, n_keep, (const int *)nullptr);
    return pngpd_launch_status();
}

int pngpd_stack_gather_lists(const int *spans, int k_views, int Pg, int G, unsigned long long seed, long long g_base,
                             int *gather, void *stream) {
    if (!spans || !gather || k_views <= 0 || k_views > 64 || Pg <= 0 || G <= 0) return PNGPD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(stack_gather_lists_kernel, dim3((Pg + 255) / 256, G), dim3(256), 0, (hipStream_t)stream, spans,
                       k_views, Pg, seed, g_base, gather);
    return pngpd_launch_status();
}

int pngpd_train_batch(const void *arena, int arena_is_f64, int P, const double *frames, const long long *labels,
                      const int *item, const int *spans, int k_views, int Pg, int *gather_ws, int G, int max_keep,
                      int N, int min_points, unsigned long long seed, long long g_base, int *counts, int *idx,
                      int *rows, unsigned char *valid, int *seg_scratch, float *out, long long *labels_out, int *n_keep,
                      void *stream) {
    if (!arena || !frames || !labels || !spans || !counts || !idx || !rows || !valid || !out || !labels_out ||
        !n_keep || P <= 0 || G <= 0 || max_keep <= 0 || N <= 0 || k_views < 0 || (k_views > 0 && (!gather_ws || Pg <= 0)))
        return PNGPD_ERR_INVALID_ARG;
    const size_t lds = (size_t)(max_keep > N ? max_keep : N) * sizeof(int);
    if (lds > 150 * 1024) return PNGPD_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int *ranges = k_views ? nullptr : spans;
    const int *gather = k_views ? gather_ws : nullptr;
    if (k_views) {
        const int rc = pngpd_stack_gather_lists(spans, k_views, Pg, G, seed ^ 0x5bd1e995a3c59ac3ull, g_base, gather_ws, stream);
        if (rc != PNGPD_OK) return rc;
    }
    // a training batch is few workgroups on 256 CUs: 1024 threads each (sanitizer builds: 256, see pngpd_common.h); long
    // gathered clouds are additionally cut into 4 segments per sample when the caller provides the scratch for it
    constexpr int CNT = PNGPD_ASAN ? 256 : 1024;
    const int SEG = (seg_scratch && k_views && Pg >= 16384 && G <= 1024) ? 4 : 1;
    int *segc = SEG > 1 ? seg_scratch : nullptr;
    for (int phase = (SEG > 1 ? 1 : 0); phase <= (SEG > 1 ? 2 : 0); ++phase) {
        if (arena_is_f64)
            hipLaunchKernelGGL((crop_count_compact_kernel<true, CNT>), dim3(G, SEG), dim3(CNT), 0, st, arena, P, frames,
                               ranges, gather, k_views ? Pg : 0, max_keep, counts, idx, item, segc, phase);
        else
            hipLaunchKernelGGL((crop_count_compact_kernel<false, CNT>), dim3(G, SEG), dim3(CNT), 0, st, arena, P, frames,
                               ranges, gather, k_views ? Pg : 0, max_keep, counts, idx, item, segc, phase);
    }
    int rc = pngpd_launch_status();
    if (rc != PNGPD_OK) return rc;
    hipLaunchKernelGGL(batch_keep_rows_kernel, dim3(1), dim3(KEEP_NT), 0, st, counts, labels, G, min_points, rows,
                       labels_out, n_keep, item);
    rc = pngpd_launch_status();
    if (rc != PNGPD_OK) return rc;
    if (arena_is_f64) {
        rc = pngpd_allow_lds((const void *)crop_resample_kernel<true>, lds);
        if (rc != PNGPD_OK) return rc;
        hipLaunchKernelGGL(crop_resample_kernel<true>, dim3(G), dim3(256), lds, st, arena, P, frames, ranges, gather,
                           k_views ? Pg : 0, counts, idx, max_keep, N, 0, min_points, seed, g_base, rows,
                           (const int *)nullptr, out, valid, item);
    } else {
        rc = pngpd_allow_lds((const void *)crop_resample_kernel<false>, lds);
        if (rc != PNGPD_OK) return rc;
        hipLaunchKernelGGL(crop_resample_kernel<false>, dim3(G), dim3(256), lds, st, arena, P, frames, ranges, gather,
                           k_views ? Pg : 0, counts, idx, max_keep, N, 0, min_points, seed, g_base, rows,
                           (const int *)nullptr, out, valid, item);
    }
    return pngpd_launch_status();
}

int pngpd_crop_count_compact_indexed(const void *cloud_sorted, int cloud_is_f64, int P, const double *spheres, int C,
                                     const double *frames, int G, int max_keep, int *counts, int *idx, void *stream) {
    if (!cloud_sorted || !spheres || !frames || !counts || !idx || P <= 0 || G <= 0 || max_keep <= 0 ||
        C != (P + 63) / 64)
        return PNGPD_ERR_INVALID_ARG;
    if (cloud_is_f64)
        hipLaunchKernelGGL(crop_count_compact_indexed_kernel<true>, dim3(G), dim3(256), 0, (hipStream_t)stream,
                           cloud_sorted, P, spheres, C, frames, max_keep, counts, idx);
    else
        hipLaunchKernelGGL(crop_count_compact_indexed_kernel<false>, dim3(G), dim3(256), 0, (hipStream_t)stream,
                           cloud_sorted, P, spheres, C, frames, max_keep, counts, idx);
    return pngpd_launch_status();
}

int pngpd_crop_indexed(const void *cloud_sorted, int cloud_is_f64, int P, const double *spheres, int C,
                       const double *frames, int G, int max_keep, int N, int mode, int min_points,
                       unsigned long long seed, long long g_base, int *counts, float *out, unsigned char *valid,
                       void *stream) {
    if (!cloud_sorted || !spheres || !frames || !counts || !out || !valid || P <= 0 || G <= 0 || max_keep <= 0 ||
        N <= 0 || (mode != 0 && mode != 1) || C != (P + 63) / 64)
        return PNGPD_ERR_INVALID_ARG;
    const size_t lds = ((size_t)(max_keep > N ? max_keep : N) + (size_t)max_keep) * sizeof(int);
    if (lds > 150 * 1024) return PNGPD_ERR_UNSUPPORTED;
    if (cloud_is_f64) {
        const int st = pngpd_allow_lds((const void *)crop_indexed_kernel<true>, lds);
        if (st != PNGPD_OK) return st;
        hipLaunchKernelGGL(crop_indexed_kernel<true>, dim3(G), dim3(256), lds, (hipStream_t)stream, cloud_sorted, P,
                           spheres, C, frames, max_keep, N, mode, min_points, seed, g_base, counts, out, valid);
    } else {
        const int st = pngpd_allow_lds((const void *)crop_indexed_kernel<false>, lds);
        if (st != PNGPD_OK) return st;
        hipLaunchKernelGGL(crop_indexed_kernel<false>, dim3(G), dim3(256), lds, (hipStream_t)stream, cloud_sorted, P,
                           spheres, C, frames, max_keep, N, mode, min_points, seed, g_base, counts, out, valid);
    }
    return pngpd_launch_status();
}

}  // extern "C"
