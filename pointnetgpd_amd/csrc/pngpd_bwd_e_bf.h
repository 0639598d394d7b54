// Pass E of the bf16 modes (round 6; included at the end of pngpd_bwd_bf.h): dz2 -> dh1 = W2^T dz2 -> g1; sums of g1,
// g1 zhat1, g1 x^T; dW2 = sum_points dz2 h1^T.
// Same outputs and phase structure as trunk_bwd_e_kernel (hand-off tiles requested a tile ahead, points double-buffered),
// with the two activation tiles in LDS as TRANSPOSED bf16:
//   dzT [128 channels][DBF_PITCH]  written 8 bytes (the lane's four consecutive points) at a time from the lane-major
//                                  hand-off values; the channel contraction W2^T dz2 reads its A operand with
//                                  ds_read_b64_tr_b16; the point contraction dW2 takes dz2 from the wave's REGISTERS;
//   h1T [64 channels][DBF_PITCH]   layer 1 (lane = channel, wave = 16 points) writes two 16-byte chunks per lane; dW2
//                                  reads its B operand as one ds_read2_b64 per k-step and block, the g1 epilogue its
//                                  ReLU mask as four ds_read_b64.
// Inherited kernel, per wave and tile (plain bf16): 56 + 24 ds_write_b32, 96 ds_read_b32 + 69 ds_read_b128 and 160
// conversions; here 8 + 2 writes, 16 transposing + 8 paired + 4 + 24 + 8 reads and 24 conversions.
#pragma once

#define EBF_W2_KS(NT) ((NT) == 3 ? 5 : 8)          // k-steps of W2^T resident in LDS (hi [+ lo] quads of both channel blocks)
#define EBF_W2_BYTES(NT) (EBF_W2_KS(NT) * 2 * ((NT) == 3 ? 2 : 1) * 64 * 16)
#define EBF_H1_HALFS (64 * DBF_PITCH)
#define EBF_LDS_BYTES(NT) ((size_t)((NT) == 3 ? 2 : 1) * (DBF_HT_HALFS + EBF_H1_HALFS) * 2 + 12 * TP * 4 + EBF_W2_BYTES(NT))

template <int NT>
__global__ __launch_bounds__(256, 2) void trunk_bwd_e_bf_kernel(
    const float *__restrict__ x, int N, const float *__restrict__ trans, TrainChan P, BwdEParams E,
    int T, int S, const f32x4 *__restrict__ z2t, const f32x4 *__restrict__ g2t, float *__restrict__ pc,
    float *__restrict__ pR, float *__restrict__ pW2, const DW3Args WT, int n_main) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x >= n_main) {   // tail workgroups (fused backward): dW3's finalize, block blockIdx.x - n_main
        dw3_finalize_body<2>(WT, (int)blockIdx.x - n_main, (double *)smem);
        return;
    }
    constexpr int NP = NT == 3 ? 2 : 1, NK = EBF_W2_KS(NT);
    u16 *dzh = (u16 *)smem;                                      // [128][DBF_PITCH]
    u16 *dzl = dzh + (NT == 3 ? DBF_HT_HALFS : 0);
    u16 *h1h = dzl + DBF_HT_HALFS;                               // [64][DBF_PITCH]
    u16 *h1l = h1h + (NT == 3 ? EBF_H1_HALFS : 0);
    float *xbuf = (float *)(h1l + EBF_H1_HALFS);                 // 2 x ([3][TP] transformed, [3][TP] original)
    f32x4 *w2l = (f32x4 *)(xbuf + 12 * TP);                      // [ks][cb1][part][lane] quads of W2^T, k-steps [0, NK)
    const Lane L;
    for (int e = L.tid; e < NK * 2 * NP * 64; e += 256) {
        const int lane = e & 63, q = e >> 6;
        const int part = q % NP, cbb = (q / NP) & 1, ks = q / (2 * NP);
        w2l[e] = ((const f32x4 *)E.w2tx)[((size_t)(cbb * 8 + ks) * 2 + part) * 64 + lane];
    }
    const int b = blockIdx.x / S, s = blockIdx.x - b * S;
    int t0, t1; tile_range(s, S, T, t0, t1);
    const float *xb = x + (size_t)b * 3 * N;
    float tm[9] = {0};
    const bool has_t = trans != nullptr;
    if (has_t) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tm[i] = trans[(size_t)b * 9 + i];
    }
    const int cb = L.wave, c2 = cb * 32 + L.j;
    const float is2 = E.is2[c2], nm2 = E.nm2[c2], a1m = E.a1m[c2], a2m = E.a2m[c2], dsc = E.dsc2[c2];
    const int pb1 = L.wave >> 1, cb1 = L.wave & 1, c1 = cb1 * 32 + L.j;
    const float w10 = P.w1[c1 * 3], w11 = P.w1[c1 * 3 + 1], w12 = P.w1[c1 * 3 + 2], bb1 = P.b1[c1];
    const float is1 = E.is1[c1], nm1 = E.nm1[c1];
    double c1d = 0.0, c2d = 0.0, r0d = 0.0, r1d = 0.0, r2d = 0.0;
    f32x16 pw0, pw1;   // dW2 rows o = cb*32 + i, columns {0,1}*32 + j
#pragma unroll
    for (int r = 0; r < 16; ++r) { pw0[r] = 0.f; pw1[r] = 0.f; }
    const L1C l1c = load_l1c(P.w1, P.b1, P.s1c, P.t1c, L);
    // lane-constant LDS offsets (halfwords)
    const int i16 = L.lane & 15, grp = (L.lane >> 4) & 1;
    const int tr_base = (8 * L.h + (i16 >> 2)) * DBF_PITCH + 32 * pb1 + 16 * grp + 4 * (i16 & 3);   // + ks*16*PITCH + t*4*PITCH
    const int wr_base = c2 * DBF_PITCH + 4 * L.h;                                                   // + 32*blk + 8*q
    const int hb0 = L.j * DBF_PITCH + 4 * L.h, hb1 = (32 + L.j) * DBF_PITCH + 4 * L.h;              // h1T rows j, 32 + j: + 32*blk + 16*s'
    const int mk_base = c1 * DBF_PITCH + 32 * pb1 + 4 * L.h;                                        // + 8*q
    constexpr int ZQ = NT == 1 ? 4 : 8;
    f32x4 gq[ZQ], zq[ZQ];
    auto fetch_tile = [&](int tile) {
        const f32x4 *gt = g2t + ((size_t)(b * T + tile) * ZQ) * 256 + L.tid;
        const f32x4 *zt = z2t + ((size_t)(b * T + tile) * ZQ) * 256 + L.tid;
#pragma unroll
        for (int i = 0; i < ZQ; ++i) { gq[i] = gt[(size_t)i * 256]; zq[i] = zt[(size_t)i * 256]; }
    };
    fetch_tile(t0);
    float px0 = 0.f, px1 = 0.f, px2 = 0.f;
    auto load_points = [&](int tile) {
        if (L.tid < TP) {
            int n = tile * TP + L.tid; n = n < N ? n : N - 1;
            px0 = xb[n]; px1 = xb[N + n]; px2 = xb[2 * N + n];
        }
    };
    auto store_points = [&](int tile) {
        if (L.tid < TP) {
            float *xs_ = xbuf + ((tile - t0) & 1) * 6 * TP, *xo_ = xs_ + 3 * TP;
            xo_[L.tid] = px0; xo_[TP + L.tid] = px1; xo_[2 * TP + L.tid] = px2;
            float y0 = px0, y1 = px1, y2 = px2;
            if (has_t) {
                y0 = fmaf(px2, tm[6], fmaf(px1, tm[3], px0 * tm[0]));
                y1 = fmaf(px2, tm[7], fmaf(px1, tm[4], px0 * tm[1]));
                y2 = fmaf(px2, tm[8], fmaf(px1, tm[5], px0 * tm[2]));
            }
            xs_[L.tid] = y0; xs_[TP + L.tid] = y1; xs_[2 * TP + L.tid] = y2;
        }
    };
    load_points(t0);
    store_points(t0);
    TM_DECL
    for (int tile = t0; tile < t1; ++tile) {
        const int nbase = tile * TP;
        float *xs = xbuf + ((tile - t0) & 1) * 6 * TP, *xo = xs + 3 * TP;
        if (tile + 1 < t1) load_points(tile + 1);
        TM(0)
        __syncthreads();   // this tile's points visible; every wave is done with the previous tile's dzT / h1T
        TM(1)
        {   // layer 1 (lane = channel, wave = 16 points) -> h1T: the lane's 16 points are two 16-byte chunks of its row
            const int p0 = L.wave * 16;
            float hv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 x0 = *(const f32x4 *)(xs + p0 + 4 * q);
                const f32x4 x1 = *(const f32x4 *)(xs + TP + p0 + 4 * q);
                const f32x4 x2 = *(const f32x4 *)(xs + 2 * TP + p0 + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {      // the fmaf chain of layer1_rows<true>: bit-identical activations
                    float z = fmaf(l1c.w2, x2[e], fmaf(l1c.w1, x1[e], fmaf(l1c.w0, x0[e], l1c.b)));
                    if (l1c.affine) z = fmaf(z, l1c.sc, l1c.sh);
                    hv[4 * q + e] = fmaxf(z, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint4 wh, wl = {0u, 0u, 0u, 0u};
                if (NT == 3) {
                    bf_split_pk2(hv[8 * u + 0], hv[8 * u + 1], wh.x, wl.x); bf_split_pk2(hv[8 * u + 2], hv[8 * u + 3], wh.y, wl.y);
                    bf_split_pk2(hv[8 * u + 4], hv[8 * u + 5], wh.z, wl.z); bf_split_pk2(hv[8 * u + 6], hv[8 * u + 7], wh.w, wl.w);
                } else {
                    wh.x = bf_pk2(hv[8 * u + 0], hv[8 * u + 1]); wh.y = bf_pk2(hv[8 * u + 2], hv[8 * u + 3]);
                    wh.z = bf_pk2(hv[8 * u + 4], hv[8 * u + 5]); wh.w = bf_pk2(hv[8 * u + 6], hv[8 * u + 7]);
                }
                *(uint4 *)(h1h + L.lane * DBF_PITCH + p0 + 8 * u) = wh;
                if (NT == 3) *(uint4 *)(h1l + L.lane * DBF_PITCH + p0 + 8 * u) = wl;
            }
        }
        TM(2)
        uint2 oh[8], ol[8];   // dz2 of (channel c2, the lane's 32 points), packed: rows of dzT AND the dW2 A operand
        {
            f32x16 a0, a1, gv0, gv1;
            if (NT == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float vz[8], vg[8];
                    bf_tile_unpack(__builtin_bit_cast(uint4, zq[i]), vz);
                    bf_tile_unpack(__builtin_bit_cast(uint4, gq[i]), vg);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (i < 2) { a0[8 * i + e] = vz[e]; gv0[8 * i + e] = vg[e]; }
                        else { a1[8 * (i - 2) + e] = vz[e]; gv1[8 * (i - 2) + e] = vg[e]; }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    gv0[r] = gq[r >> 2][r & 3]; gv1[r] = gq[(NT == 1 ? 0 : 4) + (r >> 2)][r & 3];
                    a0[r] = zq[r >> 2][r & 3]; a1[r] = zq[(NT == 1 ? 0 : 4) + (r >> 2)][r & 3];
                }
            }
            // dz2 = dsc (g2 - a1m - zhat2 a2m), zhat2 = z2 is2 + nm2 — the inherited kernel's operations (see there)
            const bool fullt = nbase + TP <= N;
            const f32x2 is22 = {is2, is2}, nm22 = {nm2, nm2}, a1m2 = {a1m, a1m}, na2m2 = {-a2m, -a2m}, dsc2 = {dsc, dsc};
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x2 dv[2];
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const int r = 4 * q + e;
                        const f32x2 zz = blk ? f32x2{a1[r], a1[r + 1]} : f32x2{a0[r], a0[r + 1]};
                        const f32x2 gg = blk ? f32x2{gv1[r], gv1[r + 1]} : f32x2{gv0[r], gv0[r + 1]};
                        const f32x2 zh = __builtin_elementwise_fma(zz, is22, nm22);
                        dv[e >> 1] = dsc2 * __builtin_elementwise_fma(zh, na2m2, gg - a1m2);
                    }
                    if (!fullt) {
                        const int row = nbase + 32 * blk + 8 * q + 4 * L.h;
                        if (row >= N) dv[0][0] = 0.f;
                        if (row + 1 >= N) dv[0][1] = 0.f;
                        if (row + 2 >= N) dv[1][0] = 0.f;
                        if (row + 3 >= N) dv[1][1] = 0.f;
                    }
                    uint2 wh, wl = {0u, 0u};
                    if (NT == 3) { bf_split_pk2(dv[0][0], dv[0][1], wh.x, wl.x); bf_split_pk2(dv[1][0], dv[1][1], wh.y, wl.y); }
                    else { wh.x = bf_pk2(dv[0][0], dv[0][1]); wh.y = bf_pk2(dv[1][0], dv[1][1]); }
                    oh[4 * blk + q] = wh; ol[4 * blk + q] = wl;
                    *(uint2 *)(dzh + wr_base + 32 * blk + 8 * q) = wh;
                    if (NT == 3) *(uint2 *)(dzl + wr_base + 32 * blk + 8 * q) = wl;
                }
            }
        }
        TM(3)
        __syncthreads();   // dzT and h1T complete
        TM(4)
        {
            // dh1[point][c1] = sum_o dz[point][o] * W2[o][c1]   (K = 128), one 32x32 tile per wave: A by transposing
            // reads of dzT, B = fragments of W2^T (LDS-resident k-steps, the rest streamed — requested first)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            constexpr int NS = 8 - NK;
            f32x4 sh_[NS > 0 ? NS : 1], sl_[NS > 0 ? NS : 1];
#pragma unroll
            for (int i = 0; i < NS; ++i) bf_wfrag<NT>(E.w2tx, 8, cb1, NK + i, L.lane, sh_[i], sl_[i]);
            const f32x4 *wq = w2l + (size_t)cb1 * NP * 64 + L.lane;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                f32x4 wh, wlo;
                if (ks < NK) {
                    wh = wq[(size_t)ks * 2 * NP * 64];
                    wlo = NT == 3 ? wq[(size_t)ks * 2 * NP * 64 + 64] : wh;
                } else { wh = sh_[ks < NK ? 0 : ks - NK]; wlo = sl_[ks < NK ? 0 : ks - NK]; }
                const u16 *p = dzh + tr_base + ks * 16 * DBF_PITCH;
                const f32x4 ah = quad_of(lds_tr16(p), lds_tr16(p + 4 * DBF_PITCH));
                f32x4 al = ah;
                if (NT == 3) {
                    const u16 *pl = dzl + tr_base + ks * 16 * DBF_PITCH;
                    al = quad_of(lds_tr16(pl), lds_tr16(pl + 4 * DBF_PITCH));
                }
                acc = bf_mma<NT>(ah, al, wh, wlo, acc);
            }
            if (tile + 1 < t1) { store_points(tile + 1); fetch_tile(tile + 1); }
            TM(5)
            // g1 = dh1 masked by ReLU(bn1) (rows past N: dz == 0 -> 0); c1 = sum g1, c2 = sum g1 zhat1, R = sum g1 x^T,
            // all on point PAIRS (packed adds / FMAs; zhat1 through the same FMA chain as layer 1)
            const f32x2 w102 = {w10, w10}, w112 = {w11, w11}, w122 = {w12, w12}, bb12 = {bb1, bb1};
            const f32x2 is12 = {is1, is1}, nm12 = {nm1, nm1};
            f32x2 cs2 = {0.f, 0.f}, cc2 = {0.f, 0.f}, rr0 = {0.f, 0.f}, rr1 = {0.f, 0.f}, rr2 = {0.f, 0.f};
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int pt = pb1 * 32 + mfma_row(4 * rq, L.lane);   // 4 consecutive points
                const f32x4 q0 = *(const f32x4 *)(xo + pt), q1 = *(const f32x4 *)(xo + TP + pt),
                            q2 = *(const f32x4 *)(xo + 2 * TP + pt);
                const f32x4 y0 = *(const f32x4 *)(xs + pt), y1 = *(const f32x4 *)(xs + TP + pt),
                            y2 = *(const f32x4 *)(xs + 2 * TP + pt);
                const uint2 hm = *(const uint2 *)(h1h + mk_base + 8 * rq);   // bf16(h1) of the four points: > 0 <=> h1 > 0
                const unsigned hw[2] = {hm.x, hm.y};
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const int r = 4 * rq + e;
                    const unsigned w = hw[e >> 1];
                    const f32x2 g = {(w & 0xffffu) ? acc[r] : 0.f, (w >> 16) ? acc[r + 1] : 0.f};
                    const f32x2 z1 = __builtin_elementwise_fma(w122, f32x2{y2[e], y2[e + 1]},
                                     __builtin_elementwise_fma(w112, f32x2{y1[e], y1[e + 1]},
                                     __builtin_elementwise_fma(w102, f32x2{y0[e], y0[e + 1]}, bb12)));
                    cs2 += g;
                    cc2 = __builtin_elementwise_fma(g, __builtin_elementwise_fma(z1, is12, nm12), cc2);
                    rr0 = __builtin_elementwise_fma(g, f32x2{q0[e], q0[e + 1]}, rr0);
                    rr1 = __builtin_elementwise_fma(g, f32x2{q1[e], q1[e + 1]}, rr1);
                    rr2 = __builtin_elementwise_fma(g, f32x2{q2[e], q2[e + 1]}, rr2);
                }
            }
            c1d += (double)(cs2[0] + cs2[1]); c2d += (double)(cc2[0] + cc2[1]);
            r0d += (double)(rr0[0] + rr0[1]); r1d += (double)(rr1[0] + rr1[1]); r2d += (double)(rr2[0] + rr2[1]);
        }
        TM(6)
        {   // dW2 += dz^T h1 over the tile's 64 points: A = the lane's own packed dz2 (registers), B = rows j and 32 + j of
            // h1T in the same point order (16 s + 4 h + {0..3}, 16 s + 8 + 4 h + {0..3})
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int blk = st >> 1, sp = st & 1, po = 32 * blk + 16 * sp;
                const f32x4 ah = quad_of(oh[4 * blk + 2 * sp], oh[4 * blk + 2 * sp + 1]);
                const f32x4 al = quad_of(ol[4 * blk + 2 * sp], ol[4 * blk + 2 * sp + 1]);
                const f32x4 b0h = quad_of(*(const uint2 *)(h1h + hb0 + po), *(const uint2 *)(h1h + hb0 + po + 8));
                const f32x4 b1h = quad_of(*(const uint2 *)(h1h + hb1 + po), *(const uint2 *)(h1h + hb1 + po + 8));
                f32x4 b0l = b0h, b1l = b1h;
                if (NT == 3) {
                    b0l = quad_of(*(const uint2 *)(h1l + hb0 + po), *(const uint2 *)(h1l + hb0 + po + 8));
                    b1l = quad_of(*(const uint2 *)(h1l + hb1 + po), *(const uint2 *)(h1l + hb1 + po + 8));
                }
                pw0 = bf_mma<NT>(ah, al, b0h, b0l, pw0);
                pw1 = bf_mma<NT>(ah, al, b1h, b1l, pw1);
            }
        }
        TM(7)
    }
    TM_END
    {
        float *oW = pW2 + (size_t)blockIdx.x * 128 * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = cb * 32 + mfma_row(r, L.lane);
            oW[o * 64 + L.j] = pw0[r];
            oW[o * 64 + 32 + L.j] = pw1[r];
        }
    }
    c1d += __shfl_xor(c1d, 32); c2d += __shfl_xor(c2d, 32);
    r0d += __shfl_xor(r0d, 32); r1d += __shfl_xor(r1d, 32); r2d += __shfl_xor(r2d, 32);
    const float c1s = (float)c1d, c2s = (float)c2d, r0 = (float)r0d, r1 = (float)r1d, r2 = (float)r2d;
    __syncthreads();   // every wave is done with dzT
    float *red = (float *)dzh;
    if (L.h == 0) {
        float *o = red + (L.wave * 32 + L.j) * 5;
        o[0] = c1s; o[1] = c2s; o[2] = r0; o[3] = r1; o[4] = r2;
    }
    __syncthreads();
    if (L.tid < 64) {
        const int cb_ = L.tid >> 5, jj = L.tid & 31;
        const float *p0 = red + ((0 * 2 + cb_) * 32 + jj) * 5;   // wave = pb1*2 + cb1
        const float *p1 = red + ((1 * 2 + cb_) * 32 + jj) * 5;
        float *oc = pc + ((size_t)blockIdx.x * 64 + L.tid) * 2;
        oc[0] = p0[0] + p1[0]; oc[1] = p0[1] + p1[1];
        float *oR = pR + ((size_t)blockIdx.x * 64 + L.tid) * 3;
        oR[0] = p0[2] + p1[2]; oR[1] = p0[3] + p1[3]; oR[2] = p0[4] + p1[4];
    }
}
