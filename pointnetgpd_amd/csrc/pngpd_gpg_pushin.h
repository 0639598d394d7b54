// libpngpd — the sampler's push-in test at candidate-generation scale (included by pngpd_gpg.hip).
//
// Reference (grasp_sampler.py:1575-1629): every potential grasp is pushed along its approach axis in S = 25 steps; at
// each step the hand is tested for collision (bottom / left / right boxes), and the backed-off, table-corrected twin of
// the step for > 10 points between the fingers and no collision; the first step whose pose collides and whose twin
// passes is the grasp (:1614-1625).  pngpd_hand_box_counts_indexed_n evaluates those 2 S poses as independent poses
// (one wave each, four exact counts per pose).  But the 2 S poses of a potential grasp share one frame — their centres
// lie on the approach axis — and the decision needs, per pose, one bit (any collision) and one saturating count.
//
// One wave per potential grasp; lane s <-> the pose of step s, lane 32 + s <-> its twin (S <= 32):
//   * the pose lanes keep the verdicts: coll (any point in a collision box) and cnt (points in the opening, twins only);
//   * broad phase: chunk spheres of gpg.CloudIndex against the hand's bounding box swept along the approach axis;
//   * per surviving chunk, first transposed (lane = pose): which poses can the chunk's sphere touch at all?  If every
//     such pose already has its verdict (collision known; count saturated above min_open or irrelevant because the twin
//     collides) the chunk is skipped without loading a point;
//   * narrow phase (lane = point): the point is transformed once into the shared frame; per box the gate axes (y, z) are
//     tested once, and the poses that hold the point are those whose centre coordinate lambda_k along the approach axis
//     lies in an interval — a short uniform loop over the poses that still need a verdict;
//   * as in the lateral sweep, this closed form is a filter: a point within a margin of any face it was tested against
//     is re-evaluated for all poses with the exact per-pose arithmetic of hand_box_counts_kernel (lane = pose).
// The result (found / sfirst) equals gpg_first_accept_kernel's on the exact counts, always.
#pragma once

namespace {

__device__ __forceinline__ unsigned long long pw_wave_or64(unsigned long long v) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) { lo |= (unsigned)__shfl_xor((int)lo, k); hi |= (unsigned)__shfl_xor((int)hi, k); }
    return ((unsigned long long)hi << 32) | lo;
}

}  // namespace

template <bool F64>
__global__ __launch_bounds__(256) void gpg_pushin_sweep_kernel(
    const void *__restrict__ cloud, int P, const double *__restrict__ spheres, int C,
    const double *__restrict__ poses2 /* (cap,S,2,12) */, const int *__restrict__ total, int cap, int S,
    const double *__restrict__ boxes, int min_open, double tol, int *__restrict__ found, int *__restrict__ sfirst,
    unsigned long long *__restrict__ stats) {
    __shared__ double bx[24];
    if (threadIdx.x < 24) bx[threadIdx.x] = boxes[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= cap) return;
    if (i >= *total) {                                      // beyond the device-side count of potential grasps
        if (lane == 0) { found[i] = 0; sfirst[i] = 0; }
        return;
    }
    // ---- setup: lane <-> pose
    const int ls = lane & 31, lk = lane >> 5;
    const bool has_pose = ls < S;
    const double *pose_l = poses2 + (((size_t)i * S + (has_pose ? ls : 0)) * 2 + lk) * 12;
    const unsigned long long fam0 = S >= 32 ? 0xFFFFFFFFull : ((1ull << S) - 1ull);
    double f0[12], lam = 0.0;
    bool slow = false, dead = false;
    {
        double fd[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) fd[j] = pose_l[j];
#pragma unroll
        for (int j = 0; j < 12; ++j) f0[j] = sw_readlane(fd[j], 0);
        // gpg_pushin_kernel parks a non-finite twin at 1e30: no point can be inside any of its boxes
        dead = has_pose && !(fabs(fd[0]) < 1e20 && fabs(fd[1]) < 1e20 && fabs(fd[2]) < 1e20);
        const double ex = fd[0] - f0[0], ey = fd[1] - f0[1], ez = fd[2] - f0[2];
        lam = f0[3] * ex + f0[4] * ey + f0[5] * ez;          // centre of pose k along the shared approach axis
        const double sy = f0[6] * ex + f0[7] * ey + f0[8] * ez, sz = f0[9] * ex + f0[10] * ey + f0[11] * ez;
        bool bad = has_pose && !dead && !(fabs(sy) < 1e-13 && fabs(sz) < 1e-13);   // centres off the approach axis
#pragma unroll
        for (int j = 3; j < 12; ++j) bad = bad || (has_pose && fd[j] != f0[j]);   // or differing axes: never, by construction
        slow = __ballot(bad) != 0ull || tol > 1e20;
    }
    const unsigned long long live_m = __ballot(has_pose && !dead);     // poses that can hold a point at all
    const unsigned long long twins = fam0 << 32;
    double lmin = (has_pose && !dead) ? lam : 1e300, lmax = (has_pose && !dead) ? lam : -1e300;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) { lmin = fmin(lmin, __shfl_xor(lmin, k)); lmax = fmax(lmax, __shfl_xor(lmax, k)); }
    const double gate_tol = 1e-11;
    const double mt = fmax(1e-11, tol * 0.005);             // margin on lambda, metres (tol: in push-in steps of 5 mm)
    // bounding box of the hand swept over [lmin, lmax] along x
    double Elo[3], Ehi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double lo = bx[2 * a], hi = bx[2 * a + 1];
#pragma unroll
        for (int b = 1; b < 4; ++b) { lo = fmin(lo, bx[b * 6 + 2 * a]); hi = fmax(hi, bx[b * 6 + 2 * a + 1]); }
        Elo[a] = lo - 1e-9 + (a == 0 ? lmin : 0.0); Ehi[a] = hi + 1e-9 + (a == 0 ? lmax : 0.0);
    }
    bool coll = false;                                      // this lane's pose: some point lies in a collision box
    int cnt = 0;                                            // this lane's pose: points in the opening (saturating use)
    unsigned long long known_coll = 0ull, sat = 0ull;       // wave-uniform views of the two
    unsigned st_pass = 0u, st_proc = 0u, st_exact = 0u;

    for (int cbase = 0; cbase < C && live_m; cbase += 64) {
        const int c = cbase + lane;
        bool pass = false;
        double g[3] = {0, 0, 0}, r = 0;
        if (c < C) {
            const double4 sp = *(const double4 *)(spheres + (size_t)c * 4);
            const double dx = sp.x - f0[0], dy = sp.y - f0[1], dz = sp.z - f0[2];
            r = sp.w * (1.0 + 1e-9) + 1e-12;
            g[0] = f0[3] * dx + f0[4] * dy + f0[5] * dz;
            g[1] = f0[6] * dx + f0[7] * dy + f0[8] * dz;
            g[2] = f0[9] * dx + f0[10] * dy + f0[11] * dz;
            pass = g[0] + r > Elo[0] && g[0] - r < Ehi[0] && g[1] + r > Elo[1] && g[1] - r < Ehi[1] &&
                   g[2] + r > Elo[2] && g[2] - r < Ehi[2];
        }
        unsigned long long work = __ballot(pass);
        st_pass += (unsigned)__popcll(work);
        while (work) {
            const int b0 = __ffsll((long long)work) - 1;
            work &= work - 1ull;
            // ---- transposed (lane = pose): which poses can this chunk's sphere touch, and do they still need it?
            const double cgx = sw_readlane(g[0], b0), cgy = sw_readlane(g[1], b0), cgz = sw_readlane(g[2], b0);
            const double cr = sw_readlane(r, b0) + 1e-9;
            bool t_coll = false, t_open = false;
            if (has_pose && !dead) {
                const double hx = cgx - lam;
#pragma unroll 1
                for (int b = 0; b < 4; ++b) {
                    const bool hit = hx + cr > bx[b * 6] && hx - cr < bx[b * 6 + 1] && cgy + cr > bx[b * 6 + 2] &&
                                     cgy - cr < bx[b * 6 + 3] && cgz + cr > bx[b * 6 + 4] && cgz - cr < bx[b * 6 + 5];
                    if (b == 0) t_open = hit; else t_coll = t_coll || hit;
                }
            }
            const unsigned long long poss_coll = __ballot(t_coll), poss_open = __ballot(t_open) & twins;
            const unsigned long long need_coll = poss_coll & ~known_coll;
            const unsigned long long need_open = poss_open & ~sat & ~known_coll;
            if (!slow && !(need_coll | need_open)) continue;
            ++st_proc;
            // ---- narrow phase (lane = point)
            const int p = (cbase + b0) * 64 + lane;
            const bool live = p < P;
            double x = 0, y = 0, z = 0;
            if (live) sw_load_point<F64>(cloud, p, x, y, z);
            bool unc = live && slow;
            unsigned long long m_coll = 0ull, m_open = 0ull;
            if (!slow) {
                const double dx = x - f0[0], dy = y - f0[1], dz = z - f0[2];
                const double hx = f0[3] * dx + f0[4] * dy + f0[5] * dz;
                const double hy = f0[6] * dx + f0[7] * dy + f0[8] * dz;
                const double hz = f0[9] * dx + f0[10] * dy + f0[11] * dz;
                asm volatile("" ::: "memory");
#pragma unroll 1
                for (int b = 0; b < 4; ++b) {
                    unsigned long long todo = b == 0 ? need_open : need_coll;       // poses that still want this box
                    if (!todo) continue;
                    // gates: y and z do not depend on the pose
                    const double dl1 = hy - bx[b * 6 + 2], dh1 = bx[b * 6 + 3] - hy;
                    const double dl2 = hz - bx[b * 6 + 4], dh2 = bx[b * 6 + 5] - hz;
                    const bool out = !live || dl1 < -gate_tol || dh1 < -gate_tol || dl2 < -gate_tol || dh2 < -gate_tol;
                    if (__ballot(!out) == 0ull) continue;
                    bool u = !out && !(dl1 > gate_tol && dh1 > gate_tol && dl2 > gate_tol && dh2 > gate_tol);
                    // x: inside pose k iff  lo < hx - lambda_k < hi
                    const double Llo = hx - bx[b * 6 + 1], Lhi = hx - bx[b * 6];
                    unsigned long long mk = 0ull;
                    while (todo) {
                        const int k = __ffsll((long long)todo) - 1;
                        todo &= todo - 1ull;
                        const double l = sw_readlane(lam, k);
                        const bool in = Llo < l && l < Lhi;
                        u = u || (!out && (fabs(l - Llo) <= mt || fabs(l - Lhi) <= mt));
                        mk |= in ? (1ull << k) : 0ull;
                    }
                    if (out) continue;
                    if (u) { unc = true; continue; }
                    if (b == 0) m_open |= mk; else m_coll |= mk;
                }
                if (unc) { m_coll = 0ull; m_open = 0ull; }
                // hand the point lanes' findings to the pose lanes
                if (__ballot((m_coll & ~known_coll) != 0ull) != 0ull) {
                    const unsigned long long nc = pw_wave_or64(m_coll);
                    coll = coll || ((nc >> lane) & 1ull);
                    known_coll |= nc;
                }
                unsigned long long oc = __ballot(m_open != 0ull) ? (pw_wave_or64(m_open) & ~sat & ~known_coll) : 0ull;
                while (oc) {
                    const int k = __ffsll((long long)oc) - 1;
                    oc &= oc - 1ull;
                    const int n = __popcll(__ballot((m_open >> k) & 1ull));
                    if (lane == k) cnt += n;
                }
            }
            // ---- exact path: lane = pose, the very arithmetic of hand_box_counts_kernel, one uncertain point a trip
            unsigned long long todo = __ballot(unc);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                ++st_exact;
                const double px = sw_readlane(x, src), py = sw_readlane(y, src), pz = sw_readlane(z, src);
                if (has_pose) {
                    double fd[12];
#pragma unroll
                    for (int j = 0; j < 12; ++j) fd[j] = pose_l[j];
                    const double dx = px - fd[0], dy = py - fd[1], dz = pz - fd[2];
                    const double gx = pn_dadd(pn_dadd(pn_dmul(fd[3], dx), pn_dmul(fd[4], dy)), pn_dmul(fd[5], dz));
                    const double gy = pn_dadd(pn_dadd(pn_dmul(fd[6], dx), pn_dmul(fd[7], dy)), pn_dmul(fd[8], dz));
                    const double gz = pn_dadd(pn_dadd(pn_dmul(fd[9], dx), pn_dmul(fd[10], dy)), pn_dmul(fd[11], dz));
                    bool in[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        in[b] = (bx[b * 6] < gx) && (bx[b * 6 + 1] > gx) && (bx[b * 6 + 2] < gy) &&
                                (bx[b * 6 + 3] > gy) && (bx[b * 6 + 4] < gz) && (bx[b * 6 + 5] > gz);
                    coll = coll || in[1] || in[2] || in[3];
                    cnt += in[0] ? 1 : 0;
                }
            }
            known_coll = __ballot(coll);
            sat = __ballot(cnt > min_open) & twins;
        }
    }
    // gpg_first_accept_kernel: the first step whose pose collides and whose twin holds > min_open points, collision-free
    const unsigned long long cm = __ballot(coll), om = __ballot(cnt > min_open);
    const unsigned acc = (unsigned)cm & (unsigned)(om >> 32) & ~(unsigned)(cm >> 32) & (unsigned)fam0;
    if (lane == 0) {
        found[i] = acc != 0u ? 1 : 0;
        sfirst[i] = acc != 0u ? __ffs((int)acc) - 1 : 0;
        if (stats) {
            atomicAdd(&stats[0], 1ull); atomicAdd(&stats[1], (unsigned long long)st_pass);
            atomicAdd(&stats[2], (unsigned long long)st_proc); atomicAdd(&stats[3], (unsigned long long)st_exact);
        }
    }
}
