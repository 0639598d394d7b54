// Round 6: passes D and E of the reduced-precision training modes (NT = 1 plain bf16, NT = 3 bf16x3) with a structure of
// their own instead of the fp32 kernels with narrower operands.  Included by pngpd_train.hip after
// BwdDParams / BwdEParams.
//
// What bounded the inherited kernels (instruction counts of a 64-point tile and wave, plain bf16, from the ISA): pass D
// 1,270 VALU + 190 LDS instructions for 30 matrix instructions — every operand element was read from an fp32 LDS tile
// and converted to bf16 AT EVERY READ (each h2 element 4 + 3 times per tile), the point contractions (Gram, dW2) read
// their operands one ds_read_b32 per element, and a v_cvt_pk_bf16_f32 was spent per VALUE.  Here:
//   * the activation tile lives in LDS ONCE, as bf16, TRANSPOSED — hT[channel][point], 144-byte rows — written with the
//     lane's own four consecutive points per ds_write_b64 (a lane of the z2t / g2t tile layout owns one channel and 32
//     points: 8 writes per lane and tile, converted in pairs, where the fp32 tile took 32 ds_write_b32);
//   * the CHANNEL contraction (h2 A, W2^T dz2) reads its A operand with ds_read_b64_tr_b16, gfx950's transposing LDS
//     read: 16 lanes fetch a [4 channels][16 points] block and every lane receives the four channels of ITS point —
//     an operand quad is two such reads, no conversion, half the bytes of the fp32 tile;
//   * the POINT contractions (Gram, dW2) take the wave's own operand straight from REGISTERS — the lane's packed
//     quads already are operand quads: lane (channel, h) holds points 16 s + 4 h + {0..3} and 16 s + 8 + 4 h + {0..3},
//     a permutation of the k index that both operands share — and the other channel blocks' from hT with one
//     ds_read2_b64 per k-step (was: 8 ds_read_b32 + 4 conversions per operand and k-step).
// Outputs (g2t tiles, pa, ps2; pc, pR, pW2) keep their layouts: the reduce / finalize kernels and pass E / the
// optimizer see no difference.  Not bit-identical to the inherited kernels (the k order inside a matrix instruction of
// the point contractions differs); same bounds in tests/test_gpu_bf16.py.
#pragma once

#define DBF_PITCH 72                        // halfwords per channel row of hT: 64 points + 8 pad = 144 B = 36 banks
#define DBF_HT_HALFS (128 * DBF_PITCH)      // one part (hi or lo) of the transposed tile: 18,432 B

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4_t;

// ds_read_b64_tr_b16 at halfword offset `off` of an LDS array: lane i of a 16-lane group supplies the address of four
// contiguous halfwords (row i/4, column chunk i%4 of a [4][16] block); it receives column i of the block, rows 0..3.
__device__ __forceinline__ uint2 lds_tr16(const u16 *p) {
    const bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t *)p);
    return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ f32x4 quad_of(uint2 a, uint2 b) {
    return f32x4{__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(b.x), __uint_as_float(b.y)};
}

// LDS of pass D (bf): hT hi [+ lo], coef row, arg row, two hit lists, hit counts
#define DBF_LDS_BYTES(NT) ((size_t)((NT) == 3 ? 2 : 1) * DBF_HT_HALFS * 2 + 1024 * 4 + 1024 * 4 + 2 * BWD_D_HITS * 8 + 64)

// Measured and rejected on this kernel (round 6, B = N = 1024, one box, A/B against the same baseline library; plain
// bf16 / bf16x3 ms per launch; this kernel 0.200 / 0.353, the inherited one 0.290 / 0.428):
//   * the cloud's hit lists bucketed by (tile, point half) ONCE per workgroup (a ballot-ranked counting sort in LDS)
//     instead of a census + barrier + compaction in every tile: 0.268 / 0.374 — the sort costs two tiles' worth of
//     serial LDS round trips per workgroup and the per-tile saving is hidden by the other resident workgroup anyway;
//   * the sparse term's W3 gathers requested ahead (1-3 sixteen-hit steps of both lists, 24-72 live registers) under
//     the h2 A contraction: 0.221-0.256 / 0.426-0.436 — the registers cost more than the latency (spills, or the
//     fragments of A moved to LDS: +50 % LDS traffic);
//   * bf16x3: the next tile's fp32 z2 a tile ahead (13 spilled registers; with the own Gram operands re-read from hT
//     instead 0.382), or the tile's z2 requested a second time for the epilogue instead of held (0.37-0.43).
template <int NT>
__global__ __launch_bounds__(256, 2) void trunk_bwd_d_bf_kernel(
    int N, TrainChan P, BwdDParams D, int T, int S, const f32x4 *__restrict__ z2t, f32x4 *__restrict__ g2t,
    float *__restrict__ pa, float *__restrict__ ps2) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16 *hTh = (u16 *)smem;                                     // [128][DBF_PITCH] bf16 hi parts of h2^T
    u16 *hTl = hTh + (NT == 3 ? DBF_HT_HALFS : 0);              // lo parts (NT == 3)
    float *cfl = (float *)(hTl + DBF_HT_HALFS);                 // [1024] coef row of this cloud
    int *idxl = (int *)(cfl + 1024);                            // [1024] arg-extremum point of every channel
    uint2 *hits = (uint2 *)(idxl + 1024);                       // [2][BWD_D_HITS] {-coef bits, (c << 5) | (point & 31)}
    int *hcnt = (int *)(hits + 2 * BWD_D_HITS);                 // [2][4]
    const Lane L;
    const int b = blockIdx.x / S, s = blockIdx.x - b * S;
    int t0, t1; tile_range(s, S, T, t0, t1);
    for (int i = L.tid; i < 1024; i += 256) {
        cfl[i] = D.coef[(size_t)b * 1024 + i];
        idxl[i] = D.idx[(size_t)b * 1024 + i];
    }
    double a1s = 0.0, a2s = 0.0;
    const int cb = L.wave;
    const int c2 = cb * 32 + L.j;
    const float sc2 = P.s2c[c2], sh2 = P.t2c[c2], is2 = D.is2[c2], nm2 = D.nm2[c2], cv = D.cvec[c2];
    const unsigned long long ltmask = (1ull << L.lane) - 1ull;
    f32x16 gm0, gm1, gm2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { gm0[r] = 0.f; gm1[r] = 0.f; gm2[r] = 0.f; }
    // fragments of A for this wave's channel block: resident for the whole kernel in plain-bf16 mode (8 quads); the
    // bf16x3 mode streams hi and lo parts from L2 per tile (its register budget holds fp32 z2 and two packed parts)
    f32x4 afh[8];
    if (NT == 1) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { f32x4 lo_; bf_wfrag<1>(D.Ax, 8, cb, ks, L.lane, afh[ks], lo_); }
    }
    // lane-constant LDS offsets (halfwords)
    const int i16 = L.lane & 15, grp = (L.lane >> 4) & 1;
    const int tr_base = (8 * L.h + (i16 >> 2)) * DBF_PITCH + 16 * grp + 4 * (i16 & 3);   // + ks*16*PITCH + t*4*PITCH + pb*32
    const int wr_base = c2 * DBF_PITCH + 4 * L.h;                                        // + 32*blk + 8*q
    const int o1 = (((cb + 1) & 3) * 32 + L.j) * DBF_PITCH + 4 * L.h;                    // + 32*blk + 16*s'
    const int o2 = (((cb + 2) & 3) * 32 + L.j) * DBF_PITCH + 4 * L.h;
    __syncthreads();   // cfl / idxl visible
    // z2 of a tile: plain bf16 keeps the NEXT tile's 16 packed registers in flight one tile ahead (0.212 -> 0.200 ms: with
    // two waves per SIMD nothing else covers the HBM round trip between a tile's first instruction and its h2 build);
    // bf16x3 (fp32 tiles, 32 registers a set) loads at the top of the tile.
    constexpr int ZQ = NT == 1 ? 4 : 8;
    f32x4 zn[ZQ];
    auto fetch_z = [&](int tile) {
        const f32x4 *zt = z2t + ((size_t)(b * T + tile) * ZQ) * 256 + L.tid;
#pragma unroll
        for (int i = 0; i < ZQ; ++i) zn[i] = zt[(size_t)i * 256];
    };
    fetch_z(t0);
    TM_DECL
    for (int tile = t0; tile < t1; ++tile) {
        const int nbase = tile * TP;
        if (NT != 1 && tile > t0) fetch_z(tile);
        f32x16 z0, z1;       // raw z2 of (this lane's rows, channel c2): live until the epilogue
        if (NT == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[8];
                bf_tile_unpack(__builtin_bit_cast(uint4, zn[i]), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { if (i < 2) z0[8 * i + e] = v[e]; else z1[8 * (i - 2) + e] = v[e]; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { z0[r] = zn[r >> 2][r & 3]; z1[r] = zn[4 + (r >> 2)][r & 3]; }
        }
        if (NT == 1 && tile + 1 < t1) fetch_z(tile + 1);
        TM(0)
        {   // hit census of this wave's channel quarter
            int clo = 0, chi = 0;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int n = idxl[L.wave * 256 + it * 64 + L.lane] - nbase;
                clo += __popcll(__ballot(n >= 0 && n < 32));
                chi += __popcll(__ballot(n >= 32 && n < TP));
            }
            if (L.lane == 0) { hcnt[L.wave] = clo; hcnt[4 + L.wave] = chi; }
        }
        TM(1)
        __syncthreads();   // counts visible; every wave is done with the previous tile's hT and hit lists
        TM(2)
        int nlo = 0, nhi = 0;
        {   // ordered compaction at the prefix offsets of the four quarters
            int olo = 0, ohi = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int a = hcnt[w], c = hcnt[4 + w];
                if (w < L.wave) { olo += a; ohi += c; }
                nlo += a; nhi += c;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int c = L.wave * 256 + it * 64 + L.lane;
                const int n = idxl[c] - nbase;
                const bool lo = n >= 0 && n < 32, hi = n >= 32 && n < TP;
                const unsigned long long mlo = __ballot(lo), mhi = __ballot(hi);
                const unsigned ncf = __float_as_uint(-cfl[c]);
                if (lo) hits[olo + __popcll(mlo & ltmask)] = uint2{ncf, (unsigned)((c << 5) | n)};
                if (hi) hits[BWD_D_HITS + ohi + __popcll(mhi & ltmask)] = uint2{ncf, (unsigned)((c << 5) | (n - 32))};
                olo += __popcll(mlo); ohi += __popcll(mhi);
            }
            if (L.tid < 24) hits[nlo + L.tid] = uint2{0u, 0u};
            else if (L.tid >= 32 && L.tid < 56) hits[BWD_D_HITS + nhi + L.tid - 32] = uint2{0u, 0u};
        }
        nlo = __builtin_amdgcn_readfirstlane(nlo);
        nhi = __builtin_amdgcn_readfirstlane(nhi);
        // h2 = relu(bn2(z2)) -> bf16 [-> residual], packed in point pairs; the lane's quad q of point block blk (points
        // 32 blk + 8 q + 4 h + {0..3}) is one 8-byte chunk of row c2 of hT — and, kept in registers, a half operand quad
        // of the Gram contraction
        uint2 oh[8], ol[8];
        const bool fullt = nbase + TP <= N;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float hv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float zz = blk ? z1[4 * q + e] : z0[4 * q + e];
                    hv[e] = fmaxf(fmaf(zz, sc2, sh2), 0.f);
                }
                if (!fullt) {      // a cloud's last, ragged tile only (wave-uniform branch)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (nbase + 32 * blk + 8 * q + 4 * L.h + e >= N) hv[e] = 0.f;
                }
                uint2 wh, wl = {0u, 0u};
                if (NT == 3) { bf_split_pk2(hv[0], hv[1], wh.x, wl.x); bf_split_pk2(hv[2], hv[3], wh.y, wl.y); }
                else { wh.x = bf_pk2(hv[0], hv[1]); wh.y = bf_pk2(hv[2], hv[3]); }
                oh[4 * blk + q] = wh; ol[4 * blk + q] = wl;
                *(uint2 *)(hTh + wr_base + 32 * blk + 8 * q) = wh;
                if (NT == 3) *(uint2 *)(hTl + wr_base + 32 * blk + 8 * q) = wl;
            }
        }
        TM(3)
        __syncthreads();   // hT and the hit lists are complete
        TM(4)
        f32x16 d0, d1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
        {   // d = h2 A: A operand by transposing reads of hT (lane = point, 8 channels), B operand = fragments of A
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                f32x4 bh, bl;
                if (NT == 1) { bh = afh[ks]; bl = bh; } else bf_wfrag<NT>(D.Ax, 8, cb, ks, L.lane, bh, bl);
                const u16 *p = hTh + tr_base + ks * 16 * DBF_PITCH;
                const f32x4 a0h = quad_of(lds_tr16(p), lds_tr16(p + 4 * DBF_PITCH));
                const f32x4 a1h = quad_of(lds_tr16(p + 32), lds_tr16(p + 4 * DBF_PITCH + 32));
                f32x4 a0l = a0h, a1l = a1h;
                if (NT == 3) {
                    const u16 *pl = hTl + tr_base + ks * 16 * DBF_PITCH;
                    a0l = quad_of(lds_tr16(pl), lds_tr16(pl + 4 * DBF_PITCH));
                    a1l = quad_of(lds_tr16(pl + 32), lds_tr16(pl + 4 * DBF_PITCH + 32));
                }
                d0 = bf_mma<NT>(a0h, a0l, bh, bl, d0);
                d1 = bf_mma<NT>(a1h, a1l, bh, bl, d1);
            }
        }
        TM(5)
        {   // sparse arg-extremum term: 16 hits per k-step; lane (j, h) supplies hits e0 + 8h .. 8h+7
            // (both lists advanced together, two gather chains in flight, 16 more live registers: 0.200 -> 0.225 ms)
            const float *w3c = D.w3 + c2;
            auto sparse = [&](const uint2 *hl, int n, f32x16 &d) {
#pragma unroll 1
                for (int e0 = 0; e0 < n; e0 += 16) {
                    float av[8], bv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const uint2 rec = hl[e0 + 8 * L.h + u];   // zero records behind the list: coef 0, row 0
                        bv[u] = w3c[(size_t)(rec.y >> 5) * 128];
                        av[u] = ((int)(rec.y & 31) == L.j) ? __uint_as_float(rec.x) : 0.f;
                    }
                    f32x4 ah, al, bh, bl;
                    bf_pack8<NT>(av, ah, al);
                    bf_pack8<NT>(bv, bh, bl);
                    d = bf_mma<NT>(ah, al, bh, bl, d);
                }
            };
            sparse(hits, nlo, d0);
            sparse(hits + BWD_D_HITS, nhi, d1);
        }
        TM(6)
        {   // Gram over the tile's points: own operand from registers, the other channel blocks' from hT.  Blocks
            // (cb,cb), (cb,cb+1) and half of (cb,cb+2) / (cb-2,cb) — the split of the inherited kernel (s2_at()).
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int blk = st >> 1, sp = st & 1;
                const bool third = (cb < 2) ? (st < 2) : (st >= 2);
                const f32x4 ah = quad_of(oh[4 * blk + 2 * sp], oh[4 * blk + 2 * sp + 1]);
                const f32x4 al = quad_of(ol[4 * blk + 2 * sp], ol[4 * blk + 2 * sp + 1]);
                gm0 = bf_mma<NT>(ah, al, ah, al, gm0);
                const int po = 32 * blk + 16 * sp;
                const f32x4 b1h = quad_of(*(const uint2 *)(hTh + o1 + po), *(const uint2 *)(hTh + o1 + po + 8));
                f32x4 b1l = b1h;
                if (NT == 3) b1l = quad_of(*(const uint2 *)(hTl + o1 + po), *(const uint2 *)(hTl + o1 + po + 8));
                gm1 = bf_mma<NT>(ah, al, b1h, b1l, gm1);
                if (third) {
                    const f32x4 b2h = quad_of(*(const uint2 *)(hTh + o2 + po), *(const uint2 *)(hTh + o2 + po + 8));
                    f32x4 b2l = b2h;
                    if (NT == 3) b2l = quad_of(*(const uint2 *)(hTl + o2 + po), *(const uint2 *)(hTl + o2 + po + 8));
                    gm2 = (cb < 2) ? bf_mma<NT>(ah, al, b2h, b2l, gm2) : bf_mma<NT>(b2h, b2l, ah, al, gm2);
                }
            }
        }
        TM(7)
        {   // epilogue: g2 = (cvec - d) masked by ReLU(bn2) and validity; running sums; lane-major hand-off
            f32x4 *gt = g2t + ((size_t)(b * T + tile) * 8) * 256 + L.tid;
            f32x16 gb0, gb1;
            const f32x2 cv2 = {cv, cv}, sc22 = {sc2, sc2}, sh22 = {sh2, sh2}, is22 = {is2, is2}, nm22 = {nm2, nm2};
            f32x2 t1s = {0.f, 0.f}, t2s = {0.f, 0.f};
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 o0, o1v;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const int r = rq * 4 + e;
                    const int row = mfma_row(r, L.lane);
                    const f32x2 zz0 = {z0[r], z0[r + 1]}, zz1 = {z1[r], z1[r + 1]};
                    f32x2 g0 = cv2 - f32x2{d0[r], d0[r + 1]}, g1 = cv2 - f32x2{d1[r], d1[r + 1]};
                    const f32x2 ac0 = __builtin_elementwise_fma(zz0, sc22, sh22), ac1 = __builtin_elementwise_fma(zz1, sc22, sh22);
                    if (fullt) {
                        g0[0] = ac0[0] > 0.f ? g0[0] : 0.f; g0[1] = ac0[1] > 0.f ? g0[1] : 0.f;
                        g1[0] = ac1[0] > 0.f ? g1[0] : 0.f; g1[1] = ac1[1] > 0.f ? g1[1] : 0.f;
                    } else {
                        g0[0] = (nbase + row < N && ac0[0] > 0.f) ? g0[0] : 0.f;
                        g0[1] = (nbase + row + 1 < N && ac0[1] > 0.f) ? g0[1] : 0.f;
                        g1[0] = (nbase + 32 + row < N && ac1[0] > 0.f) ? g1[0] : 0.f;
                        g1[1] = (nbase + 33 + row < N && ac1[1] > 0.f) ? g1[1] : 0.f;
                    }
                    t1s += g0 + g1;
                    t2s = __builtin_elementwise_fma(g0, __builtin_elementwise_fma(zz0, is22, nm22),
                          __builtin_elementwise_fma(g1, __builtin_elementwise_fma(zz1, is22, nm22), t2s));
                    o0[e] = g0[0]; o0[e + 1] = g0[1]; o1v[e] = g1[0]; o1v[e + 1] = g1[1];
                    if (NT == 1) { gb0[r] = g0[0]; gb0[r + 1] = g0[1]; gb1[r] = g1[0]; gb1[r + 1] = g1[1]; }
                }
                if (NT != 1) {
                    gt[(size_t)rq * 256] = o0;
                    gt[(size_t)(4 + rq) * 256] = o1v;
                }
            }
            a1s += (double)(t1s[0] + t1s[1]);
            a2s += (double)(t2s[0] + t2s[1]);
            if constexpr (NT == 1) bf_tile_store((uint4 *)g2t + ((size_t)(b * T + tile) * 4) * 256 + L.tid, gb0, gb1);
        }
        TM(8)
    }
    TM_END
    a1s += __shfl_xor(a1s, 32);
    a2s += __shfl_xor(a2s, 32);
    if (L.h == 0) {
        float *o = pa + ((size_t)blockIdx.x * 128 + c2) * 2;
        o[0] = (float)a1s; o[1] = (float)a2s;
    }
    {
        float *o = ps2 + ((size_t)blockIdx.x * 12 + cb * 3) * 1024 + L.lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o[r * 64] = gm0[r];
            o[1024 + r * 64] = gm1[r];
            o[2048 + r * 64] = gm2[r];
        }
    }
}

#include "pngpd_bwd_e_bf.h"
