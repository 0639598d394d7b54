// EXPERIMENT (round 5): the software-pipelined bf16 / bf16x3 inference trunk of round 4
// (tools/probes/trunk_infer_x3_pipelined.inc: layer 1 of tile t+1 rides in the layer-3 stream of tile t, two barriers per
// tile instead of three) with the ONE change the verdict's hypothesis asks for: the layer-1 pieces no longer fetch their
// wave-uniform weights with scalar loads — which share the LGKM counter with layer 3's A-fragment reads AND return out
// of order, so the s_waitcnt in front of every piece was lgkmcnt(0), a full drain of the LDS queue — but with
// ds_read_b128 broadcasts from a 1 KB LDS table (in-order, so a counted wait suffices).
//
// MEASURED AND REJECTED (profiles/r05_ab_x3p.txt; alternating processes on one box, tools/bench_eval_bf.py, B = N = 1024):
//   lock-step product kernel   bf16x3 1.2868 / 1.2875 / 1.2893 ms   bf16 0.5051 / 0.5057 / 0.5059 ms
//   this kernel                bf16x3 1.2959 / 1.2987 / 1.3003 ms   bf16 0.5104 / 0.5104 / 0.5109 ms     (0.7 % / 1.0 % SLOWER)
// with all 36 tests of test_gpu_bf16 / test_gpu_infer_x3 / test_gpu_refine green on it (bit-identical results).  So the
// scalar loads were not what kept round 4's pipelined kernel from gaining: with them gone the layer-1 pieces still cost
// the layer-3 stream as much as the phase they removed.  254 VGPRs (bf16x3; 207 in the lock-step kernel), 62 SGPRs.
//
// Not part of the build.  To rebuild the experiment: include this file in pngpd_trunk_infer_x3.hip in front of
// launch_infer_bf and, for !ARG, launch trunk_infer_x3p_kernel<NT, XBF> with X3_LDS_BYTES + 1024 bytes of dynamic LDS
// (same arguments), e.g. under -DPNGPD_X3_PIPELINED into build_probe/lib_x3p.so, and point PNGPD_LIB at it.
#pragma once

template <int NT, bool XBF>
__global__ __launch_bounds__(512, 2) void trunk_infer_x3p_kernel(
    const void *__restrict__ x, int N, const float *__restrict__ trans,
    const float *__restrict__ w1, const float *__restrict__ b1,
    const u16 *__restrict__ w2x, const float *__restrict__ b2,
    const u16 *__restrict__ w3x, const float *__restrict__ b3,
    int relu_last, int T, int S, float *__restrict__ out, int *__restrict__ out_arg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16 *h1h = (u16 *)smem_raw;                 // [XP][X1S]
    u16 *h1l = h1h + XP * X1S;
    u16 *h2h = h1l + XP * X1S;                  // [XP][X2S]
    u16 *h2l = h2h + XP * X2S;
    float *xs = (float *)(h2l + XP * X2S);      // [3][XP]
    float *rm = xs + 3 * XP;                    // [1024]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / S, s = blockIdx.x - b * S;
    const int t0 = (int)(((long)s * T) / S), t1 = (int)(((long)(s + 1) * T) / S);
    const size_t xo = (size_t)b * 3 * N;   // element offset of this cloud
    float tm[9] = {0};
    const bool has_t = trans != nullptr;
    if (has_t) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tm[i] = trans[(size_t)b * 9 + i];
    }
    for (int i = tid; i < 1024; i += 512) rm[i] = -INFINITY;
    // layer-1 constants (w1 row + bias of every channel) in LDS: the pipelined layer-1 pieces read them with in-order
    // ds_read_b128 broadcasts instead of scalar loads (which return out of order and force s_waitcnt lgkmcnt(0))
    f32x4 *l1c = (f32x4 *)(rm + 1024);
    if (tid < 64) { f32x4 v; v[0] = w1[tid * 3]; v[1] = w1[tid * 3 + 1]; v[2] = w1[tid * 3 + 2]; v[3] = b1[tid]; l1c[tid] = v; }
    __shared__ int s_bad;   // non-finite input coordinate seen: poison the pooled row (see trunk_infer_kernel)
    if (tid == 0) s_bad = 0;
    __syncthreads();

    // layer-3 weight fragments (hi+lo of one 32-channel block = 64 VGPRs)
    f32x4 wah[8], wal[8];
    // NT == 1 (plain bf16): the wave's FOUR channel blocks stay resident in 128 VGPRs for the whole kernel.  With the
    // matrix cores 16x faster than in fp32 the weight stream itself was the bottleneck: every wave re-fetched 8 KB per
    // block and tile from L2 (2 KB per point and workgroup; with all 256 CUs on the same 256 KB of weights: half of each
    // XCD's L2 bandwidth), and tools/phase_times_x3.py showed 2,150 cycles per block waiting for the fragments to land —
    // 35 % of the kernel.  (NT == 3 would need 256 VGPRs for hi + lo of four blocks and keeps streaming.)
    constexpr bool WRES = NT == 1;   // (the arg search needs the registers: the ARG variant streams its weights)
    f32x4 wres[WRES ? 4 : 1][8];
    if (WRES) {
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            const f32x4 *p = (const f32x4 *)w3x + (size_t)((wave + 8 * ci) * 8) * 2 * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) wres[ci][ks] = p[(ks * 2) * 64];
        }
    }

    float px0 = 0.f, px1 = 0.f, px2 = 0.f;
    // ---- software pipeline over the tiles (round 4): layer 1 of tile t+1 rides in the layer-3 stream of tile t -------
    // The lock-step kernel walked  stage -> B -> layer 1 -> B -> layer 2 -> B -> layer 3  per tile, and 27-29 % of a
    // tile passed with the matrix pipe idle (tools/phase_times_x3.py).  Layer 1 writes the h1 tile, which layer 3 never
    // reads, and on the bf16 matrix pipe VALU work DOES overlap (tools/probes/mfma_valu_overlap.hip): every thread now
    // fetches ITS OWN point of the next tile straight from the cloud (no staged xs, no barrier), evaluates its 16
    // layer-1 channels in four pieces BETWEEN the four layer-3 channel blocks of the current tile and writes them to h1.
    // Per tile: layer 2 -> B -> layer 3 (+ layer 1 of the next tile) -> B: two barriers instead of three, and staging +
    // layer 1 are off the critical path.  Results are bit-identical to the lock-step kernel (same operations per value).
    const int p1 = tid & 127, g1 = wave >> 1;        // layer 1: thread = (point p1 of the tile, 16-channel group g1)
    auto fetch_point = [&](int tile) {
        int n = tile * XP + p1; n = n < N ? n : N - 1;
        px0 = ldx<XBF>(x, xo + n); px1 = ldx<XBF>(x, xo + N + n); px2 = ldx<XBF>(x, xo + 2 * (size_t)N + n);
    };
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;              // the (transformed) point layer 1 is working on
    auto take_point = [&]() {
        q0 = px0; q1 = px1; q2 = px2;
        if (has_t) {
            q0 = fmaf(px2, tm[6], fmaf(px1, tm[3], px0 * tm[0]));
            q1 = fmaf(px2, tm[7], fmaf(px1, tm[4], px0 * tm[1]));
            q2 = fmaf(px2, tm[8], fmaf(px1, tm[5], px0 * tm[2]));
        }
        if (!__builtin_isfinite(px0 + px1 + px2)) s_bad = 1;
    };
    auto layer1_piece = [&](int piece) {             // channels g1*16 + 4*piece .. +3 of point p1 -> h1 (hi [, lo])
        u16 hv[4], lv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = g1 * 16 + piece * 4 + e;   // wave-uniform: one ds_read_b128 broadcast per channel
            const f32x4 wc = l1c[c];
            const float z = fmaxf(fmaf(wc[2], q2, fmaf(wc[1], q1, fmaf(wc[0], q0, wc[3]))), 0.f);
            split2(z, hv[e], lv[e]);
        }
        uint2 vh, vl;
        vh.x = hv[0] | ((unsigned)hv[1] << 16); vh.y = hv[2] | ((unsigned)hv[3] << 16);
        vl.x = lv[0] | ((unsigned)lv[1] << 16); vl.y = lv[2] | ((unsigned)lv[3] << 16);
        *(uint2 *)(h1h + p1 * X1S + g1 * 16 + piece * 4) = vh;
        if (NT == 3) *(uint2 *)(h1l + p1 * X1S + g1 * 16 + piece * 4) = vl;
    };
    fetch_point(t0);
    take_point();
#pragma unroll
    for (int piece = 0; piece < 4; ++piece) layer1_piece(piece);      // prologue: layer 1 of the first tile
    if (t0 + 1 < t1) fetch_point(t0 + 1);

    TM_DECL
    for (int tile = t0; tile < t1; ++tile) {
        TM(0)
        __syncthreads();   // h1 of this tile is complete; every wave is done with layer 3 of the previous tile (h2)
        TM(1)
        {   // layer 2 (64 -> 128): wave owns channel block cb = wave & 3 and point blocks 2q, 2q+1
            const int cb = wave & 3, pb0 = (wave >> 2) * 2;
            f32x4 w2h[4], w2l[4];
            load_wx<4, NT>(w2h, w2l, w2x, cb, lane);
            f32x16 a0 = {0}, a1 = {0};
            const int r0 = (pb0 * 32 + j) * X1S + h * 8, r1 = ((pb0 + 1) * 32 + j) * X1S + h * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 ah0 = *(const f32x4 *)(h1h + r0 + ks * 16), ah1 = *(const f32x4 *)(h1h + r1 + ks * 16);
                a0 = mfma_bf(ah0, w2h[ks], a0); a1 = mfma_bf(ah1, w2h[ks], a1);
                if (NT == 3) {
                    const f32x4 al0 = *(const f32x4 *)(h1l + r0 + ks * 16), al1 = *(const f32x4 *)(h1l + r1 + ks * 16);
                    a0 = mfma_bf(ah0, w2l[ks], a0); a1 = mfma_bf(ah1, w2l[ks], a1);
                    a0 = mfma_bf(al0, w2h[ks], a0); a1 = mfma_bf(al1, w2h[ks], a1);
                }
            }
            const float bias = b2[cb * 32 + j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma_row(r, lane);
                u16 hi, lo;
                split2(fmaxf(a0[r] + bias, 0.f), hi, lo);
                h2h[(pb0 * 32 + row) * X2S + cb * 32 + j] = hi;
                if (NT == 3) h2l[(pb0 * 32 + row) * X2S + cb * 32 + j] = lo;
                split2(fmaxf(a1[r] + bias, 0.f), hi, lo);
                h2h[((pb0 + 1) * 32 + row) * X2S + cb * 32 + j] = hi;
                if (NT == 3) h2l[((pb0 + 1) * 32 + row) * X2S + cb * 32 + j] = lo;
            }
        }
        TM(2)
        __syncthreads();   // h2 complete; every wave is done reading h1 (layer 2): layer 1 of the next tile may write it
        TM(3)
        const bool more = tile + 1 < t1;   // workgroup-uniform
        if (more) {
            take_point();
            if (tile + 2 < t1) fetch_point(tile + 2);   // in flight during the whole of this tile's layer 3
        }
        auto block4 = [&](int cb, const f32x4 (&wah)[8], const f32x4 (&wal)[8]) {
            const float rmc = rm[cb * 32 + j];   // requested before the block's MFMAs, merged after them
            f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
            const int ro = j * X2S + h * 8;
            f32x4 ah[4], al[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ah[q] = *(const f32x4 *)(h2h + ro + q * 32 * X2S);
                al[q] = (NT == 3) ? *(const f32x4 *)(h2l + ro + q * 32 * X2S) : ah[q];
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                f32x4 nh[4], nl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    nh[q] = ah[q]; nl[q] = al[q];
                    if (ks < 7) {
                        nh[q] = *(const f32x4 *)(h2h + ro + q * 32 * X2S + (ks + 1) * 16);
                        if (NT == 3) nl[q] = *(const f32x4 *)(h2l + ro + q * 32 * X2S + (ks + 1) * 16);
                    }
                }
                c0 = mfma_bf(ah[0], wah[ks], c0); c1 = mfma_bf(ah[1], wah[ks], c1);
                c2 = mfma_bf(ah[2], wah[ks], c2); c3 = mfma_bf(ah[3], wah[ks], c3);
                if (NT == 3) {
                    c0 = mfma_bf(ah[0], wal[ks], c0); c1 = mfma_bf(ah[1], wal[ks], c1);
                    c2 = mfma_bf(ah[2], wal[ks], c2); c3 = mfma_bf(ah[3], wal[ks], c3);
                    c0 = mfma_bf(al[0], wah[ks], c0); c1 = mfma_bf(al[1], wah[ks], c1);
                    c2 = mfma_bf(al[2], wah[ks], c2); c3 = mfma_bf(al[3], wah[ks], c3);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { ah[q] = nh[q]; al[q] = nl[q]; }
            }
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) m = max3f(m, max3f(c0[r], c1[r], c2[r]), c3[r]);
            float mlo, mhi;
            half_pair(m, mlo, mhi);   // v_permlane32_swap: the LDS is this kernel's scarcest resource, no ds_bpermute
            if (h == 0) rm[cb * 32 + j] = fmaxf(rmc, fmaxf(mlo, mhi));
        };
        if constexpr (WRES) {
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                if (more) layer1_piece(ci);
                block4(wave + 8 * ci, wres[ci], wres[ci]);
            }
        } else {
#pragma unroll 1
            for (int ci = 0; ci < 4; ++ci) {
                load_wx<8, NT>(wah, wal, w3x, wave + 8 * ci, lane);
                if (more) layer1_piece(ci);
                block4(wave + 8 * ci, wah, wal);
            }
        }
        TM(7)
    }
    TM_END_TO(pngpd_tm_x3)
    if (h == 0) {
        float *o = out + ((size_t)b * S + s) * 1024;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            const int c = (wave + 8 * ci) * 32 + j;
            float v = rm[c] + b3[c];
            if (relu_last) v = fmaxf(v, 0.f);
            o[c] = s_bad ? __builtin_nanf("") : v;
        }
    }
}


