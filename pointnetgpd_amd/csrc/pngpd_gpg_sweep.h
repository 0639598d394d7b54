// libpngpd — the GPG sampler's lateral sweep at candidate-generation scale (BASELINE configs[4] upstream).
//
// Reference (dex-net/src/dexnet/grasping/grasp_sampler.py): for every sample point and each of the R = 19 rotations
// about the minor axis (:1524-1529) the hand is tried at D = 21 lateral offsets dy (:1531-1563) — four
// check_collision_square passes per pose (:336-393) — and the MIDDLE offset among those with an empty bottom / left /
// right box and a non-empty opening is kept (:1565-1567), subject to the 30-degree rule (:1570-1573).
//
// pngpd_hand_box_counts* evaluates those R x D poses as independent poses (one wave each).  But the D poses of a
// rotation share one frame — only the centre moves, along the binormal — and the selection needs no COUNTS, only two
// predicates per pose (opening non-empty, any collision).  This kernel therefore works per (sample point, rotation):
//
//   * ONE wave per unit; the cloud is the Morton-ordered cloud + chunk spheres of gpg.CloudIndex;
//   * broad phase: chunk spheres against the hand's bounding box SWEPT over the D offsets (lane = chunk);
//   * narrow phase: a point is transformed ONCE into the unit's frame; per box the set of offsets d whose pose holds
//     the point is an integer interval, computed in closed form (the pose centres are affine in d) and OR-ed into two
//     D-bit masks per lane (opening, collision) — no ballots, no per-pose loop;
//   * the closed form is a FILTER, not the decision: wherever an interval end lies within a margin of an integer (or a
//     gate coordinate within a margin of a face) the point is "uncertain" and is re-evaluated for all D poses with the
//     exact per-pose arithmetic of hand_box_counts_kernel (lane = offset) — so the masks equal the ones derived from
//     the exact counts, always (tests compare them; `tol` = 1e30 forces the exact path everywhere);
//   * saturation: a chunk's sphere bounds the offsets its points can reach per box (a conservative bit mask); once the
//     masks accumulated so far already hold every bit a chunk could add — or, for the opening, the bit is dead because
//     that offset already collides — the chunk is skipped without loading a point.  On a dense cloud the masks
//     saturate after a few dozen chunks; skipped chunks can only set bits that are set, so the result is unchanged;
//   * the select step (:1565-1573, gpg_select_kernel) is fused: flag / dsel leave directly — and a unit whose 30-degree
//     rule (:1570-1573; a function of the approach axis and the pose centre only) fails at EVERY offset can never yield
//     a potential grasp: it leaves before touching the cloud (the reference sweeps first and tests afterwards; the
//     result is the same).  On a table-top scene that is every sample point whose approach axis is not pointing
//     downwards by more than 30 degrees — most of them.
#pragma once
// (included by pngpd_gpg.hip, which sets `#pragma clang fp contract(off)` for the whole translation unit)

namespace {

template <bool F64>
__device__ __forceinline__ void sw_load_point(const void *__restrict__ cloud, int p, double &x, double &y, double &z) {
    if (F64) {
        const double *c = (const double *)cloud + (size_t)p * 3;
        x = c[0]; y = c[1]; z = c[2];
    } else {
        const float *c = (const float *)cloud + (size_t)p * 3;
        x = (double)c[0]; y = (double)c[1]; z = (double)c[2];
    }
}

__device__ __forceinline__ double sw_readlane(double v, int lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)__double2loint(v), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)__double2hiint(v), lane);
    return __hiloint2double((int)hi, (int)lo);
}

__device__ __forceinline__ unsigned sw_wave_or(unsigned v) {
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) v |= (unsigned)__shfl_xor((int)v, k);
    return v;
}

}  // namespace

// poses (LR*D,12) and ab (LR,6) as written by gpg_enumerate_kernel; boxes (4,6) = opening, left, right, bottom.
// flag / dsel (LR): gpg_select_kernel's outputs (dsel is meaningful where flag is set).  masks (LR,2) or NULL: the
// opening / collision bit masks (bit d) of EVERY unit (no pruning then).
template <bool F64>
__global__ __launch_bounds__(256, 4) void gpg_sweep_select_kernel(
    const void *__restrict__ cloud, int P, const double *__restrict__ spheres, int C,
    const double *__restrict__ poses, const double *__restrict__ ab, int LR, int D,
    const double *__restrict__ boxes, const double *__restrict__ prm, double tol, int *__restrict__ flag,
    int *__restrict__ dsel, unsigned *__restrict__ masks, unsigned long long *__restrict__ stats) {
    __shared__ double bx[24];                               // the four boxes: uniform reads (LDS broadcast), no registers
    if (threadIdx.x < 24) bx[threadIdx.x] = boxes[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= LR) return;                                    // wave-uniform; no further barriers in this kernel
    // ---- per-unit setup: lane d (< D) looks at pose d, every lane keeps the reference pose (d = D/2)
    const int dref = D >> 1;
    const double *pose_d = poses + ((size_t)t * D + (lane < D ? lane : dref)) * 12;
    double f0[12], s0[3], w[3];
    unsigned rule_m;
    bool slow = false;                                      // the centres are not affine in d (never, for enumerate's poses)
    {
        double fd[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) fd[i] = pose_d[i];
        // the 30-degree rule of gpg_select_kernel, evaluated at every offset (the selected one is among them)
        const double paz = ab[(size_t)t * 6 + 2];
        rule_m = (unsigned)__ballot(lane < D && pn_dadd(fd[2], pn_dmul(paz, prm[1])) < pn_dsub(fd[2], prm[2]));
        if (rule_m == 0u && !masks) {
            if (lane == 0) { flag[t] = 0; dsel[t] = 0; }
            return;
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) f0[i] = sw_readlane(fd[i], dref);
        // s_d = R (c_d - c_ref): where pose d's centre sits in the reference pose's frame
        const double ex = fd[0] - f0[0], ey = fd[1] - f0[1], ez = fd[2] - f0[2];
        double sd[3];
        sd[0] = f0[3] * ex + f0[4] * ey + f0[5] * ez;
        sd[1] = f0[6] * ex + f0[7] * ey + f0[8] * ez;
        sd[2] = f0[9] * ex + f0[10] * ey + f0[11] * ez;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s0[a] = sw_readlane(sd[a], 0);
            const double sl = sw_readlane(sd[a], D - 1);
            w[a] = D > 1 ? (sl - s0[a]) / (double)(D - 1) : 0.0;
            const double dev = fabs(sd[a] - (s0[a] + (double)lane * w[a]));
            slow = slow || (lane < D && !(dev < 1e-13));
        }
        // the three axes of a pose must also be THE SAME for every d (bit for bit, as enumerate writes them)
#pragma unroll
        for (int i = 3; i < 12; ++i) slow = slow || (lane < D && fd[i] != f0[i]);
        slow = __ballot(slow) != 0ull || tol > 1e20;        // tol > 1e20: force the exact path everywhere (tests)
    }
    // axis a of a box test:  lo < h_a - d w_a < hi.  |w_a| tiny (the frame axis is orthogonal to the sweep): a GATE,
    // the same verdict for every d; otherwise an open interval of offsets, with the margin that covers the closed
    // form's rounding (h: |h| < 2; the affine model's residual < 1e-13) expressed in offsets.
    bool gate[3];
    double inv_w[3], err[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        gate[a] = fabs(w[a]) < 1e-13;                       // |d w| < 32e-13: inside the gate tolerance below
        inv_w[a] = gate[a] ? 0.0 : 1.0 / w[a];
        err[a] = gate[a] ? 0.0 : tol + (4e-15 + 1e-13) * fabs(inv_w[a]);
    }
    const double gate_tol = fmax(1e-11, tol * 1e-3);
    // swept bounding box of the whole hand in h-coordinates (h = R(p - c_ref) - s0; pose d sees h - d w)
    double Elo[3], Ehi[3], swlo[3], swhi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double lo = bx[2 * a], hi = bx[2 * a + 1];
#pragma unroll
        for (int b = 1; b < 4; ++b) { lo = fmin(lo, bx[b * 6 + 2 * a]); hi = fmax(hi, bx[b * 6 + 2 * a + 1]); }
        const double sweep = (double)(D - 1) * w[a];
        swlo[a] = fmin(0.0, sweep) - 1e-9; swhi[a] = fmax(0.0, sweep) + 1e-9;
        Elo[a] = lo + swlo[a]; Ehi[a] = hi + swhi[a];
    }
    unsigned known_open = 0u, known_coll = 0u;               // wave-uniform: the masks accumulated so far
    unsigned st_pass = 0u, st_proc = 0u, st_exact = 0u;      // diagnostics (stats != NULL): chunks passed / evaluated, exact trips
    const unsigned full = D >= 32 ? 0xFFFFFFFFu : ((1u << D) - 1u);

    for (int cbase = 0; cbase < C; cbase += 64) {
        const int c = cbase + lane;
        bool pass = false;
        unsigned bmask = 0u, poss_open = 0u, poss_coll = 0u;
        if (c < C) {
            const double4 sp = *(const double4 *)(spheres + (size_t)c * 4);
            const double dx = sp.x - f0[0], dy = sp.y - f0[1], dz = sp.z - f0[2];
            const double r = sp.w * (1.0 + 1e-9) + 1e-12;
            double g[3];
            g[0] = f0[3] * dx + f0[4] * dy + f0[5] * dz - s0[0];
            g[1] = f0[6] * dx + f0[7] * dy + f0[8] * dz - s0[1];
            g[2] = f0[9] * dx + f0[10] * dy + f0[11] * dz - s0[2];
            pass = g[0] + r > Elo[0] && g[0] - r < Ehi[0] && g[1] + r > Elo[1] && g[1] - r < Ehi[1] &&
                   g[2] + r > Elo[2] && g[2] - r < Ehi[2];
            if (pass) {
                double gp[3], gm[3];                        // sphere extent, the sweep folded in: compare with lo / hi as is
#pragma unroll
                for (int a = 0; a < 3; ++a) { gp[a] = g[a] + r - swlo[a]; gm[a] = g[a] - r - swhi[a]; }
#pragma unroll 1
                for (int b = 0; b < 4; ++b) {
                    const bool hit = gp[0] > bx[b * 6] && gm[0] < bx[b * 6 + 1] && gp[1] > bx[b * 6 + 2] &&
                                     gm[1] < bx[b * 6 + 3] && gp[2] > bx[b * 6 + 4] && gm[2] < bx[b * 6 + 5];
                    if (!hit) continue;
                    bmask |= 1u << b;
                    // offsets at which a point of this sphere can be inside box b (conservative): for every interval
                    // axis,  tau w in (g - r - hi, g + r - lo)
                    double tl = -1.0, th = (double)D;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        if (gate[a]) continue;
                        const double t1 = (g[a] - r - bx[b * 6 + 2 * a + 1]) * inv_w[a];
                        const double t2 = (g[a] + r - bx[b * 6 + 2 * a]) * inv_w[a];
                        tl = fmax(tl, fmin(t1, t2) - err[a] - 1e-6);
                        th = fmin(th, fmax(t1, t2) + err[a] + 1e-6);
                    }
                    int dlo = (int)ceil(tl), dhi = (int)floor(th);      // the integers inside the (already padded) range
                    dlo = dlo < 0 ? 0 : dlo; dhi = dhi > D - 1 ? D - 1 : dhi;
                    const unsigned pm = dhi < dlo ? 0u : (((dhi >= 31 ? 0xFFFFFFFFu : ((2u << dhi) - 1u)) & ~((1u << dlo) - 1u)) & full);
                    if (b == 0) poss_open |= pm; else poss_coll |= pm;
                }
                pass = bmask != 0u;
            }
        }
        unsigned long long work = __ballot(pass);
        st_pass += (unsigned)__popcll(work);
        // Walk the surviving chunks; a chunk is evaluated only if it can still add a bit (see "saturation" above).  The
        // chunk loop is a chain of dependent loads, so the next chunk's points are requested before the current ones are
        // evaluated (with the masks known at that moment: a prefetched chunk may turn out to be unnecessary).
        auto next_needed = [&](unsigned long long &wk) {
            while (wk) {
                const int b = __ffsll((long long)wk) - 1;
                const unsigned po = (unsigned)__builtin_amdgcn_readlane((int)poss_open, b);
                const unsigned pc = (unsigned)__builtin_amdgcn_readlane((int)poss_coll, b);
                const unsigned dead = masks ? 0u : known_coll;      // (the debug masks want every opening bit)
                if (slow || (po & ~known_open & ~dead) || (pc & ~known_coll)) return b;
                wk &= wk - 1ull;
            }
            return -1;
        };
        double nx = 0, ny = 0, nz = 0;
        int nb = next_needed(work);
        if (nb >= 0 && (cbase + nb) * 64 + lane < P) sw_load_point<F64>(cloud, (cbase + nb) * 64 + lane, nx, ny, nz);
        while (nb >= 0) {
            const int b0 = nb;
            ++st_proc;
            work &= work - 1ull;                            // b0 is the lowest set bit of `work`
            const unsigned bm = (unsigned)__builtin_amdgcn_readlane((int)bmask, b0);
            const bool live = (cbase + b0) * 64 + lane < P;
            const double x = nx, y = ny, z = nz;
            nb = next_needed(work);
            if (nb >= 0 && (cbase + nb) * 64 + lane < P) sw_load_point<F64>(cloud, (cbase + nb) * 64 + lane, nx, ny, nz);
            bool unc = live && slow;
            if (!slow) {
                const double dx = x - f0[0], dy = y - f0[1], dz = z - f0[2];
                double h[3];
                h[0] = f0[3] * dx + f0[4] * dy + f0[5] * dz - s0[0];
                h[1] = f0[6] * dx + f0[7] * dy + f0[8] * dz - s0[1];
                h[2] = f0[9] * dx + f0[10] * dy + f0[11] * dz - s0[2];
                unsigned m_open = 0u, m_coll = 0u;
                asm volatile("" ::: "memory");              // box bounds are re-read from LDS here, not kept in 48 registers
#pragma unroll 1
                for (int b = 0; b < 4; ++b) {
                    if (!(bm & (1u << b))) continue;        // wave-uniform: the chunk's sphere misses box b's sweep
                    // gate axes first: a handful of compares decide most lanes (and often the whole wave)
                    bool none = !live, u = false;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        if (!gate[a]) continue;
                        const double dl = h[a] - bx[b * 6 + 2 * a], dh = bx[b * 6 + 2 * a + 1] - h[a];
                        const bool out = dl < -gate_tol || dh < -gate_tol;
                        none = none || out;
                        u = u || (!out && !(dl > gate_tol && dh > gate_tol));
                    }
                    if (__ballot(!none) == 0ull) continue;  // wave-uniform: nobody can be inside box b at any offset
                    if (none) continue;
                    // interval axes: offsets tau with lo < h - tau w < hi
                    double tl = -4.0, th = (double)D + 3.0, margin = 0.0;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        if (gate[a]) continue;
                        const double t1 = (h[a] - bx[b * 6 + 2 * a]) * inv_w[a], t2 = (h[a] - bx[b * 6 + 2 * a + 1]) * inv_w[a];
                        tl = fmax(tl, fmin(t1, t2));
                        th = fmin(th, fmax(t1, t2));
                        margin = fmax(margin, err[a]);
                    }
                    tl = fmin(tl, (double)D + 2.0); th = fmax(th, -3.0);      // keep the int conversions in range
                    // integers d with tl < d < th; an end within `margin` of an in-range integer decides that integer:
                    // such a point goes to the exact path
                    const double fl = floor(tl), fh = floor(th);
                    const double frl = tl - fl, frh = th - fh;
                    const double kl = frl <= margin ? fl : fl + 1.0, kh = frh <= margin ? fh : fh + 1.0;
                    u = u || margin >= 0.25 ||
                        ((frl <= margin || frl >= 1.0 - margin) && kl >= 0.0 && kl <= (double)(D - 1) && th > kl - margin) ||
                        ((frh <= margin || frh >= 1.0 - margin) && kh >= 0.0 && kh <= (double)(D - 1) && tl < kh + margin);
                    if (u) { unc = true; continue; }
                    int dmin = (int)fl + 1, dmax = frh == 0.0 ? (int)fh - 1 : (int)fh;
                    dmin = dmin < 0 ? 0 : dmin;
                    dmax = dmax > D - 1 ? D - 1 : dmax;
                    if (dmax < dmin) continue;
                    const unsigned m = ((dmax >= 31 ? 0xFFFFFFFFu : ((2u << dmax) - 1u)) & ~((1u << dmin) - 1u)) & full;
                    if (b == 0) m_open |= m; else m_coll |= m;
                }
                if (unc) { m_open = 0u; m_coll = 0u; }
                // fold this chunk's new bits into the wave-uniform masks (only when some lane has one)
                if (__ballot((m_open & ~known_open) | (m_coll & ~known_coll)) != 0ull) {
                    known_open |= sw_wave_or(m_open);
                    known_coll |= sw_wave_or(m_coll);
                }
            }
            // ---- exact path: lane = offset d, the very arithmetic of hand_box_counts_kernel, one uncertain point a trip
            unsigned long long todo = __ballot(unc);
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                ++st_exact;
                const double px = sw_readlane(x, src), py = sw_readlane(y, src), pz = sw_readlane(z, src);
                bool in[4] = {false, false, false, false};
                if (lane < D) {
                    double fd[12];                          // pose d again (L2-resident): the exact path is rare
#pragma unroll
                    for (int i = 0; i < 12; ++i) fd[i] = pose_d[i];
                    const double dx = px - fd[0], dy = py - fd[1], dz = pz - fd[2];
                    const double gx = pn_dadd(pn_dadd(pn_dmul(fd[3], dx), pn_dmul(fd[4], dy)), pn_dmul(fd[5], dz));
                    const double gy = pn_dadd(pn_dadd(pn_dmul(fd[6], dx), pn_dmul(fd[7], dy)), pn_dmul(fd[8], dz));
                    const double gz = pn_dadd(pn_dadd(pn_dmul(fd[9], dx), pn_dmul(fd[10], dy)), pn_dmul(fd[11], dz));
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        in[b] = (bx[b * 6] < gx) && (bx[b * 6 + 1] > gx) && (bx[b * 6 + 2] < gy) &&
                                (bx[b * 6 + 3] > gy) && (bx[b * 6 + 4] < gz) && (bx[b * 6 + 5] > gz);
                }
                known_open |= (unsigned)__ballot(in[0]);
                known_coll |= (unsigned)(__ballot(in[1]) | __ballot(in[2]) | __ballot(in[3]));
            }
        }
    }
    const unsigned open_m = known_open & full, coll_m = known_coll & full;
    if (stats && lane == 0) {
        atomicAdd(&stats[0], 1ull); atomicAdd(&stats[1], (unsigned long long)st_pass);
        atomicAdd(&stats[2], (unsigned long long)st_proc); atomicAdd(&stats[3], (unsigned long long)st_exact);
    }
    if (lane == 0) {
        if (masks) { masks[(size_t)t * 2] = open_m; masks[(size_t)t * 2 + 1] = coll_m; }
        // gpg_select_kernel: the middle admissible offset (:1565-1567), then the 30-degree rule at it (:1570-1573)
        const unsigned ok = open_m & ~coll_m;
        const int n_ok = __popc(ok);
        int f = 0, ds = 0;
        if (n_ok > 0) {
            const int target = (n_ok + 1) / 2 - 1;
            unsigned rest = ok;
            for (int k = 0; k < target; ++k) rest &= rest - 1u;
            ds = __ffs((int)rest) - 1;
            f = (int)((rule_m >> ds) & 1u);
        }
        flag[t] = f; dsel[t] = ds;
    }
}
