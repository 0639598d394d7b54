// libpngpd — the sampler's neighbourhood moments over the spatial index (included by pngpd_gpg.hip).
//
// grasp_sampler.py:1471-1485: the (at most) 100 nearest cloud points within the r-ball of a sample point and
// M = sum n n^T over them.  gpg_normal_moments_kernel finds them by a radix select over ALL P points (11 passes over the
// cloud per sample point).  On a dense cloud the 100 nearest points live in a handful of the index's 64-point chunks:
//   1. bound: walking the chunk spheres by increasing FARTHEST distance ub = |q - c| + r until they hold >= max_nn
//      points gives U = the last ub; at least max_nn points lie within U, so (when U < r_ball) the max_nn nearest do too;
//   2. candidates: chunks whose NEAREST distance lb = |q - c| - r is <= min(U, r_ball) — in chunk order, so that every
//      sum below has a fixed order (deterministic results);
//   3. the selection of gpg_normal_moments_kernel, unchanged (radix select on the bit pattern of d^2, ties at the cut
//      towards the lower ORIGINAL index, M accumulated in fp64), over the candidate chunks' points only.
// Same selected set, and the additions of M are re-ordered into the whole-cloud kernel's order: M is bit-identical.
#pragma once

constexpr int GPG_MAXCAND = 2048;      // candidate chunk ids kept in LDS; more -> every chunk is a candidate
constexpr int GPG_MAXSEL = 1024;       // selected points staged in LDS for the ordered sum (max_nn <= GPG_MAXSEL)

template <bool F64>
__global__ __launch_bounds__(256) void gpg_normal_moments_indexed_kernel(
    const void *__restrict__ cloud /* Morton-sorted */, const int *__restrict__ order /* sorted pos -> original index */,
    const double *__restrict__ normals /* original order */, int P, const double *__restrict__ spheres, int C,
    const double *__restrict__ queries, double radius, int max_nn, double *__restrict__ M_out,
    int *__restrict__ nsel_out) {
    __shared__ int shi[4];
    __shared__ double shd[4 * 6];
    __shared__ int hist[256];
    __shared__ int pick[2];
    __shared__ int cand[GPG_MAXCAND];
    __shared__ int wcnt[4];
    __shared__ double red_v[4];
    __shared__ int red_i[4];
    __shared__ int sel_raw[GPG_MAXSEL], sel_sorted[GPG_MAXSEL];
    __shared__ int sel_n;
    __shared__ int sel_state[2];       // [0] points held by the chunks picked so far, [1] last picked chunk
    __shared__ double sel_ub;
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double qx = queries[s * 3 + 0], qy = queries[s * 3 + 1], qz = queries[s * 3 + 2];
    const double r2 = radius * radius;

    auto sphere_bounds = [&](int c, double &lb, double &ub) {
        const double4 sp = *(const double4 *)(spheres + (size_t)c * 4);
        const double dx = sp.x - qx, dy = sp.y - qy, dz = sp.z - qz;
        const double d = sqrt(dx * dx + dy * dy + dz * dz);
        const double r = sp.w * (1.0 + 1e-9) + 1e-12;       // conservative against rounding, as in the sweep kernels
        ub = (d + r) * (1.0 + 1e-12);
        lb = fmax(0.0, (d - r) * (1.0 - 1e-12) - 1e-15);
    };
    // ---- 1. U: chunks by increasing ub until they hold max_nn points (lexicographic (ub, c) order, block-wide minimum)
    if (tid == 0) { sel_state[0] = 0; sel_state[1] = -1; sel_ub = -1.0; }
    __syncthreads();
    double U = 0.0;
    for (int round = 0; round < C; ++round) {
        const double last_ub = sel_ub; const int last_c = sel_state[1];
        double best = 1e300; int best_c = 0x7fffffff;
        for (int c = tid; c < C; c += 256) {
            double lb, ub;
            sphere_bounds(c, lb, ub);
            const bool after = ub > last_ub || (ub == last_ub && c > last_c);
            if (after && (ub < best || (ub == best && c < best_c))) { best = ub; best_c = c; }
        }
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) {
            const double ov = __shfl_xor(best, k); const int oc = __shfl_xor(best_c, k);
            if (ov < best || (ov == best && oc < best_c)) { best = ov; best_c = oc; }
        }
        if (lane == 0) { red_v[wave] = best; red_i[wave] = best_c; }
        __syncthreads();
        if (tid == 0) {
            double b = red_v[0]; int bc = red_i[0];
            for (int w = 1; w < 4; ++w) if (red_v[w] < b || (red_v[w] == b && red_i[w] < bc)) { b = red_v[w]; bc = red_i[w]; }
            if (bc != 0x7fffffff) {
                sel_ub = b; sel_state[1] = bc;
                sel_state[0] += (bc == C - 1) ? P - 64 * (C - 1) : 64;
            } else {
                sel_state[0] = 0x7fffffff;                  // every chunk picked: the whole cloud is the candidate set
            }
        }
        __syncthreads();
        U = sel_ub;
        if (sel_state[0] >= max_nn) break;
    }
    const bool all_points = sel_state[0] == 0x7fffffff;      // fewer than max_nn points in the whole cloud
    const double bound = all_points ? radius : fmin(U, radius);
    // ---- 2. candidate chunks (lb <= bound and lb < r_ball), in chunk order
    int ncand = 0;
    bool use_all = false;
    for (int base = 0; base < C; base += 256) {
        const int c = base + tid;
        bool keep = false;
        if (c < C) {
            double lb, ub;
            sphere_bounds(c, lb, ub);
            keep = lb <= bound && lb < radius;
        }
        const unsigned long long m = __ballot(keep);
        __syncthreads();
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int n = wcnt[w]; if (w < wave) woff += n; tot += n; }
        if (keep) {
            const int pos = ncand + woff + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < GPG_MAXCAND) cand[pos] = c;
        }
        ncand += tot;
    }
    __syncthreads();
    if (ncand > GPG_MAXCAND) { use_all = true; ncand = C; }
    const int npts = ncand * 64;
    // point i of the candidate domain -> (valid, coordinates, ORIGINAL index)
    auto point_at = [&](int i, double &x, double &y, double &z, int &orig) {
        const int c = use_all ? (i >> 6) : cand[i >> 6];
        const int ps = c * 64 + (i & 63);
        if (i >= npts || ps >= P) return false;
        gpg_load_point<F64>(cloud, ps, x, y, z);
        orig = order[ps];
        return true;
    };
    // ---- 3. the selection of gpg_normal_moments_kernel over the candidate points
    auto count_le_bits = [&](unsigned long long T) {         // #points with d2 < r2 and bits(d2) <= T
        int cnt = 0;
        for (int i = tid; i < npts; i += 256) {
            double x, y, z; int o;
            if (!point_at(i, x, y, z, o)) continue;
            const double d2 = gpg_dist2(x, y, z, qx, qy, qz);
            cnt += (d2 < r2 && (unsigned long long)__double_as_longlong(d2) <= T) ? 1 : 0;
        }
        return block_sum_int(cnt, shi);
    };
    const unsigned long long r2bits = (unsigned long long)__double_as_longlong(r2);
    const int in_ball = count_le_bits(r2bits);
    unsigned long long T = r2bits;
    int tie_keep = 0x7fffffff;
    if (in_ball > max_nn) {
        unsigned long long lo = 0;
        int remaining = max_nn;
        for (int pos = 7; pos >= 0; --pos) {
            hist[tid] = 0;
            __syncthreads();
            const int sh_hi = 8 * (pos + 1);
            for (int i = tid; i < npts; i += 256) {
                double x, y, z; int o;
                if (!point_at(i, x, y, z, o)) continue;
                const double d2 = gpg_dist2(x, y, z, qx, qy, qz);
                const unsigned long long key = (unsigned long long)__double_as_longlong(d2);
                const bool match = pos == 7 ? true : (key >> sh_hi) == lo;
                if (d2 < r2 && match) atomicAdd(&hist[(int)((key >> (8 * pos)) & 255ull)], 1);
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
        if (n_le > max_nn) {                                 // ties at the cut: keep the lowest ORIGINAL indices
            const int n_lt = T ? count_le_bits(T - 1) : 0;
            const int need = max_nn - n_lt;
            int ilo = 0, ihi = P - 1;
            while (ilo < ihi) {
                const int imid = ilo + ((ihi - ilo) >> 1);
                int cnt = 0;
                for (int i = tid; i < npts; i += 256) {
                    double x, y, z; int o;
                    if (!point_at(i, x, y, z, o)) continue;
                    const double d2 = gpg_dist2(x, y, z, qx, qy, qz);
                    cnt += (o <= imid && d2 < r2 && (unsigned long long)__double_as_longlong(d2) == T) ? 1 : 0;
                }
                if (block_sum_int(cnt, shi) >= need) ihi = imid; else ilo = imid + 1;
            }
            tie_keep = ilo;
        }
    }
    // The selected points (<= max_nn) are summed in EXACTLY the order of gpg_normal_moments_kernel — thread (p mod 256)
    // adds its points by increasing ORIGINAL index p, then the same butterfly — so that M is bit-identical to the
    // whole-cloud kernel's: np.linalg.eig's eigenvector signs (the host half, :1493) can flip on a last-bit change of M.
    double m[6] = {0, 0, 0, 0, 0, 0};
    int nsel = 0;
    if (tid == 0) sel_n = 0;
    __syncthreads();
    for (int i = tid; i < npts; i += 256) {
        double x, y, z; int p;
        if (!point_at(i, x, y, z, p)) continue;
        const double d2 = gpg_dist2(x, y, z, qx, qy, qz);
        const unsigned long long b = (unsigned long long)__double_as_longlong(d2);
        const bool sel = d2 < r2 && (b < T || (b == T && p <= tie_keep));
        if (!sel) continue;
        ++nsel;
        if (d2 == 0.0) continue;                             // :1477 skips the sample point itself
        const int slot = atomicAdd(&sel_n, 1);
        if (slot < GPG_MAXSEL) sel_raw[slot] = p;
    }
    __syncthreads();
    const int n_add = sel_n < GPG_MAXSEL ? sel_n : GPG_MAXSEL;   // (the host wrapper keeps max_nn <= GPG_MAXSEL)
    for (int e = tid; e < n_add; e += 256) {                 // rank sort by original index (indices are distinct)
        const int mine = sel_raw[e];
        int rank = 0;
        for (int j = 0; j < n_add; ++j) rank += sel_raw[j] < mine ? 1 : 0;
        sel_sorted[rank] = mine;
    }
    __syncthreads();
    for (int j = 0; j < n_add; ++j) {
        const int p = sel_sorted[j];
        if ((p & 255) != tid) continue;
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
