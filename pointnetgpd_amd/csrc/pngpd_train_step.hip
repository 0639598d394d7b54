// libpngpd — one C-ABI entry per direction of the training graph: the whole forward / backward of a trunk
// (conv1..3 + bn1..3 [+ ReLU] + max-pool, PointNetGPD/model/pointnet.py:29-33 and :140-149 in train mode) and of an
// FC head (fc1/bn/relu, fc2/bn/relu, fc3 + tail, pointnet.py:35-43 and :191-194) is enqueued by ONE call with
// caller-provided workspace.  These entries only SEQUENCE the pass kernels of pngpd_train.hip / pngpd_train_glue.hip
// (the same launches, in the same order, with the same arithmetic as calling the per-pass entry points one by one —
// tests compare the two bit for bit), so that an eager training step of the reference's own recipe (main_1v.py:72-76,
// batch 64 x 750 points) costs eight foreign-function calls instead of ~140.
#include "pngpd_internal.h"
#include "pngpd_glue_bodies.h"

namespace {

// Bump allocator over a caller-provided buffer; base == nullptr is a dry run that only measures.
struct Carve {
    char *base; size_t cap, off; bool ok;
    Carve(void *b, size_t c) : base((char *)b), cap(c), off(0), ok(true) {}
    template <typename T> T *take(size_t n) {
        const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        T *p = base ? (T *)(base + off) : nullptr;
        off += bytes;
        if (base && off > cap) ok = false;
        return p;
    }
};

struct TrunkDims {
    int B, N, S, Sc, T, blk, nt, nt_side, cpr, R;
    size_t tile_floats;   // fp32 words of one z2 / g2 tile buffer in the side passes' storage type
};

bool trunk_dims(const pngpd_trunk_train_t *a, TrunkDims &d) {
    if (!a || a->B <= 0 || a->N <= 0) return false;
    if (a->precision != 0 && a->precision != 1 && a->precision != 3) return false;
    if (a->refine < 0 || a->refine > 2) return false;
    d.B = a->B; d.N = a->N; d.S = a->S;
    d.T = (a->N + 63) / 64;
    if (d.S < 1 || d.S > d.T) return false;
    d.blk = d.B * d.S;
    d.nt = a->precision;
    d.nt_side = (a->precision && !a->fp32_side) ? a->precision : 0;
    const int T128 = (a->N + 127) / 128;
    d.Sc = d.nt ? (d.S < T128 ? d.S : T128) : d.S;
    if (d.Sc < 1) d.Sc = 1;
    d.cpr = d.B / 16; if (d.cpr > 16) d.cpr = 16; if (d.cpr < 1) d.cpr = 1;
    d.R = (d.B + d.cpr - 1) / d.cpr;
    const size_t full = pngpd_trunk_g2t_bytes(d.B, d.N) / 4;
    d.tile_floats = d.nt_side == 1 ? full / 2 : full;
    return true;
}

// What the forward leaves for the backward (and the operands both directions share).
struct TrunkSave {
    double *mom, *stats1, *stats2, *stats3, *sh;
    float *chan1, *chan2, *w2p, *w2tp, *z2t;
    unsigned short *w2x, *w2tx;
};

bool carve_save(Carve &c, const TrunkDims &d, TrunkSave &s) {
    s.mom = c.take<double>((size_t)d.B * 9);
    s.stats1 = c.take<double>(140);
    s.stats2 = c.take<double>(256);
    s.stats3 = c.take<double>(2048);
    s.sh = c.take<double>(128);
    s.chan1 = c.take<float>(4 * 64);
    s.chan2 = c.take<float>(4 * 128);
    s.w2p = c.take<float>(128 * 64);
    s.w2tp = c.take<float>(128 * 64);
    s.w2x = c.take<unsigned short>(2 * 128 * 64);
    s.w2tx = c.take<unsigned short>(2 * 128 * 64);
    s.z2t = c.take<float>(d.tile_floats);
    return c.ok;
}

struct TrunkFwdScratch { float *part, *pmax, *psum, *psh, *w3sp, *zex; int *parg; unsigned short *w3sx; };

bool carve_fwd(Carve &c, const TrunkDims &d, TrunkFwdScratch &f) {
    f.part = c.take<float>((size_t)d.blk * 256);
    f.pmax = c.take<float>((size_t)d.B * d.Sc * 1024);
    f.parg = c.take<int>((size_t)d.B * d.Sc * 1024);
    f.psum = c.take<float>((size_t)d.B * d.Sc * 2048);
    f.psh = c.take<float>((size_t)d.B * d.Sc * (d.nt ? 2 : 1) * 128);
    f.w3sp = c.take<float>(1024 * 128);
    f.w3sx = c.take<unsigned short>(2 * 1024 * 128);
    f.zex = c.take<float>((size_t)d.B * 1024);
    return c.ok;
}

struct TrunkBwdScratch {
    float *coef, *Ap, *cvec, *Gp, *g2t, *pa, *ps2, *evec, *pc, *pR, *pW2;
    double *m12, *G, *S2c, *c12, *Rb;
    unsigned short *Ax;
};

bool carve_bwd(Carve &c, const TrunkDims &d, TrunkBwdScratch &w) {
    w.coef = c.take<float>((size_t)d.B * 1024);
    w.m12 = c.take<double>(2048);
    w.Ap = c.take<float>(128 * 128);
    w.cvec = c.take<float>(128);
    w.Ax = c.take<unsigned short>(2 * 128 * 128);
    w.Gp = c.take<float>((size_t)d.R * 1024 * 128);
    w.g2t = c.take<float>(d.tile_floats);
    w.pa = c.take<float>((size_t)d.blk * 256);
    w.ps2 = c.take<float>((size_t)d.blk * 12 * 1024);
    w.G = c.take<double>(1024 * 128);
    w.S2c = c.take<double>(12 * 1024);
    w.evec = c.take<float>(3 * 128);
    w.pc = c.take<float>((size_t)d.blk * 128);
    w.pR = c.take<float>((size_t)d.blk * 192);
    w.pW2 = c.take<float>((size_t)d.blk * 128 * 64);
    w.c12 = c.take<double>(128);
    w.Rb = c.take<double>((size_t)d.B * 192);
    return c.ok;
}

#define CHK(call) do { int st__ = (call); if (st__ != PNGPD_OK) return st__; } while (0)

}  // namespace

extern "C" {

size_t pngpd_struct_bytes(int which) {   /* 0: pngpd_trunk_train_t, 1: pngpd_head_train_t, 2: pngpd_fold_model_t — binding self-check */
    return which == 0 ? sizeof(pngpd_trunk_train_t) : which == 1 ? sizeof(pngpd_head_train_t)
         : which == 2 ? sizeof(pngpd_fold_model_t) : 0;
}

size_t pngpd_trunk_train_save_bytes(const pngpd_trunk_train_t *a) {
    TrunkDims d; TrunkSave s;
    if (!trunk_dims(a, d)) return 0;
    Carve c(nullptr, 0);
    carve_save(c, d, s);
    return c.off;
}

size_t pngpd_trunk_train_scratch_bytes(const pngpd_trunk_train_t *a) {
    TrunkDims d;
    if (!trunk_dims(a, d)) return 0;
    Carve cf(nullptr, 0), cb(nullptr, 0);
    TrunkFwdScratch f; TrunkBwdScratch w;
    carve_fwd(cf, d, f);
    carve_bwd(cb, d, w);
    return cf.off > cb.off ? cf.off : cb.off;
}

int pngpd_trunk_train_fwd(const pngpd_trunk_train_t *a, void *stream) {
    TrunkDims d;
    if (!trunk_dims(a, d)) return PNGPD_ERR_INVALID_ARG;
    if (!a->x || !a->w1 || !a->b1 || !a->g1 || !a->be1 || !a->w2 || !a->b2 || !a->g2 || !a->be2 || !a->w3 ||
        !a->b3 || !a->g3 || !a->be3 || !a->pooled || !a->idx || !a->zhat || !a->save || !a->scratch)
        return PNGPD_ERR_INVALID_ARG;
    Carve cs(a->save, a->save_bytes), cf(a->scratch, a->scratch_bytes);
    TrunkSave s; TrunkFwdScratch f;
    if (!carve_save(cs, d, s) || !carve_fwd(cf, d, f)) return PNGPD_ERR_WORKSPACE;
    const int B = d.B, N = d.N, S = d.S;
    const float *x = a->x, *T = a->trans;
    // ---- pass A (per-cloud moments) and every weight re-layout of the step in ONE launch: the two are independent
    {
        PackArgs P{}; int n = 0;
        P.mom_x = x; P.mom = s.mom; P.mom_N = N; P.mom_B = B;
        const bool need_f32_side = d.nt_side == 0;
        const bool refine = d.nt && a->refine;
        if (need_f32_side || refine) P.job[n++] = PackJob{a->w2, nullptr, s.w2p, 128, 64, 0, 0, 0};
        if (need_f32_side && a->need_bwd) P.job[n++] = PackJob{a->w2, nullptr, s.w2tp, 64, 128, 1, 0, 0};
        if (d.nt) {
            P.job[n++] = PackJob{a->w2, nullptr, s.w2x, 128, 64, 0, 0, 1};
            P.job[n++] = PackJob{a->w3, a->g3, f.w3sx, 1024, 128, 0, 0, 1};
            if (d.nt_side && a->need_bwd) P.job[n++] = PackJob{a->w2, nullptr, s.w2tx, 64, 128, 1, 0, 1};
            if (a->refine) P.job[n++] = PackJob{a->w3, a->g3, f.w3sp, 1024, 128, 0, 0, 0};   // (2: the VALU variant reads its rows from the fragments)
        } else {
            P.job[n++] = PackJob{a->w3, a->g3, f.w3sp, 1024, 128, 0, 0, 0};
        }
        CHK(pngpd_train_pack_launch(P, n, stream));
    }
    // ---- BN1 (closed form from the per-cloud moments)
    CHK(pngpd_bn1_finalize(s.mom, T, B, N, a->w1, a->b1, a->g1, a->be1, a->eps, a->momentum, a->rm1, a->rv1,
                           a->nbt1, s.chan1, s.stats1, stream));
    const float *s1c = s.chan1, *t1c = s.chan1 + 64;
    // ---- pass B (z2 = W2 h1 once, stored for passes C / D / E) + BN2
    const bool store_z2 = d.nt_side ? true : (d.nt == 0 || a->need_bwd);
    if (d.nt_side)
        CHK(pngpd_trunk_bn2_stats_bf(x, B, N, T, a->w1, a->b1, s1c, t1c, s.w2x, d.nt_side, S, f.part, s.z2t, stream));
    else
        CHK(pngpd_trunk_bn2_stats(x, B, N, T, a->w1, a->b1, s1c, t1c, s.w2p, S, f.part, store_z2 ? s.z2t : nullptr,
                                  stream));
    {
        RFArgs A; A.M = (double)B * N; A.eps = (double)a->eps; A.momentum = (double)a->momentum;
        RFSeg g{}; g.in = f.part; g.outer = 1; g.R = d.blk; g.n = 256; g.kind = RF_BN2;
        g.p0 = a->b2; g.p1 = a->g2; g.p2 = a->be2; g.rm = a->rm2; g.rv = a->rv2; g.nbt = a->nbt2;
        g.f0 = s.chan2; g.s0 = s.stats2;
        A.seg[0] = g;
        CHK(pngpd_reduce_fin_launch(A, 1, stream));
    }
    const float *s2c = s.chan2, *t2c = s.chan2 + 128;
    // ---- pass C + BN3
    int prows;
    if (d.nt) {
        CHK(pngpd_trunk_fwd_train_bf(x, B, N, T, a->w1, a->b1, s1c, t1c, s.w2x, s2c, t2c, f.w3sx, d.nt, d.Sc, f.pmax,
                                     f.parg, f.psum, f.psh, d.nt_side == d.nt ? s.z2t : nullptr, stream));
        prows = B * d.Sc * 2;
    } else {
        CHK(pngpd_trunk_fwd_train(x, B, N, T, a->w1, a->b1, s1c, t1c, s.w2p, s2c, t2c, f.w3sp, S, f.pmax, f.parg,
                                  f.psum, f.psh, s.z2t, stream));
        prows = B * S;
    }
    {
        RFArgs A; A.M = (double)B * N; A.eps = (double)a->eps; A.momentum = (double)a->momentum;
        RFSeg g{}; g.in = f.psum; g.outer = 1; g.R = B * d.Sc; g.n = 1024; g.kind = RF_BN3;
        g.p0 = a->b3; g.p1 = a->g3; g.rm = a->rm3; g.rv = a->rv3; g.nbt = a->nbt3; g.s0 = s.stats3;
        A.seg[0] = g;
        RFSeg h{}; h.in = f.psh; h.out = s.sh; h.outer = 1; h.R = prows; h.n = 128; h.kind = RF_F64;
        A.seg[1] = h;
        CHK(pngpd_reduce_fin_launch(A, 2, stream));
    }
    CHK(pngpd_pool_finalize(f.pmax, f.parg, B, d.Sc, s.stats3, a->g3, a->be3, a->eps, a->relu_last, a->pooled,
                            a->idx, a->zhat, stream));
    if (d.nt && a->refine) {
        // the reduced-precision pass C chose the points; their values are re-evaluated in exact fp32 (layers 1-2 for
        // the B*1024 arg-max points + one 128-long contraction each) and the pooled outputs rebuilt from those
        CHK(pngpd_trunk_pool_refine(x, B, N, T, a->w1, a->b1, s1c, t1c, s.w2p, s2c, t2c,
                                    f.w3sp, a->w3, a->g3, a->idx, d.cpr,
                                    a->refine == 1 ? 0 : PNGPD_REFINE_VALU_VARIANT, f.zex, stream));
        CHK(pngpd_pool_finalize(f.zex, a->idx, B, 1, s.stats3, a->g3, a->be3, a->eps, a->relu_last, a->pooled, a->idx,
                                a->zhat, stream));
    }
    return PNGPD_OK;
}

int pngpd_trunk_train_bwd(const pngpd_trunk_train_t *a, void *stream) {
    TrunkDims d;
    if (!trunk_dims(a, d)) return PNGPD_ERR_INVALID_ARG;
    if (!a->x || !a->w1 || !a->b1 || !a->g1 || !a->w2 || !a->g2 || !a->w3 || !a->g3 || !a->pooled || !a->idx ||
        !a->zhat || !a->dp || !a->dW1 || !a->dg1 || !a->dbe1 || !a->dW2 || !a->dg2 || !a->dbe2 || !a->dW3 ||
        !a->dg3 || !a->dbe3 || !a->save || !a->scratch)
        return PNGPD_ERR_INVALID_ARG;
    if (a->dT && !a->trans) return PNGPD_ERR_INVALID_ARG;
    Carve cs(a->save, a->save_bytes), cb(a->scratch, a->scratch_bytes);
    TrunkSave s; TrunkBwdScratch w;
    if (!carve_save(cs, d, s) || !carve_bwd(cb, d, w)) return PNGPD_ERR_WORKSPACE;
    const int B = d.B, N = d.N, S = d.S, nt = d.nt_side;
    const float *x = a->x, *T = a->trans;
    const float *s1c = s.chan1, *t1c = s.chan1 + 64, *is1 = s.chan1 + 128, *nm1 = s.chan1 + 192;
    const float *s2c = s.chan2, *t2c = s.chan2 + 128, *is2 = s.chan2 + 256, *nm2 = s.chan2 + 384;
    // ---- BN3 affine grads, sparse-term weights, dense-correction scalars, pass-D operands
    CHK(pngpd_bn3_bwd_prep(a->dp, a->pooled, a->zhat, B, N, a->g3, s.stats3, a->eps, a->relu_last, w.coef, a->dg3,
                           a->dbe3, w.m12, stream));
    // ---- arg-extremum gather (sparse term of dW3) with A / cvec's finalize as tail workgroups of the same launch (the
    //      two are independent; pass D, one launch later, is their first reader), then pass D
    const ACvecArgs AT{a->w3, a->g3, s.stats3, w.m12, s.sh, (double)B * N, (double)a->eps, w.Ap, w.cvec};
    if (nt) {
        CHK(pngpd_bwd_gather_impl(x, B, N, T, a->w1, a->b1, s1c, t1c, nullptr, s.w2x, nt, s2c, t2c, a->idx, w.coef, d.cpr,
                                  w.Gp, &AT, stream));
        PackArgs P{};
        P.job[0] = PackJob{w.Ap, nullptr, w.Ax, 128, 128, 0, 1, 1};
        CHK(pngpd_train_pack_launch(P, 1, stream));
        CHK(pngpd_trunk_bwd_d_bf(x, B, N, s2c, t2c, is2, nm2, w.Ax, nt, w.cvec, a->w3, a->idx, w.coef, s.z2t, S, w.g2t,
                                 w.pa, w.ps2, stream));
    } else {
        CHK(pngpd_bwd_gather_impl(x, B, N, T, a->w1, a->b1, s1c, t1c, s.w2p, nullptr, 0, s2c, t2c, a->idx, w.coef, d.cpr,
                                  w.Gp, &AT, stream));
        CHK(pngpd_trunk_bwd_d(x, B, N, T, a->w1, a->b1, s1c, t1c, s.w2p, s2c, t2c, is2, nm2, w.Ap, w.cvec, a->w3,
                              a->idx, w.coef, s.z2t, S, w.g2t, w.pa, w.ps2, stream));
    }
    {
        RFArgs A; A.M = (double)B * N; A.eps = (double)a->eps; A.momentum = 0.0;
        RFSeg g{}; g.in = w.Gp; g.out = w.G; g.outer = 1; g.R = d.R; g.n = 1024 * 128; g.kind = RF_F64;
        RFSeg q{}; q.in = w.ps2; q.out = w.S2c; q.outer = 1; q.R = d.blk; q.n = 12 * 1024; q.kind = RF_F64;
        RFSeg e{}; e.in = w.pa; e.outer = 1; e.R = d.blk; e.n = 256; e.kind = RF_EPREP;
        e.p0 = a->g2; e.d0 = s.stats2; e.f0 = a->dg2; e.f1 = a->dbe2; e.f2 = w.evec;
        A.seg[0] = g; A.seg[1] = e; A.seg[2] = q;
        CHK(pngpd_reduce_fin_launch(A, 3, stream));
    }
    // ---- pass E (also contracts dW2 = sum_points dz2 h1^T on the MFMA), with dW3's finalize as tail workgroups: it
    //      reads only what the reduction above wrote, and nothing in this entry reads dW3
    const DW3Args WT{w.G, w.S2c, s.sh, (double)B * N, a->w3, a->g3, s.stats3, w.m12, (double)a->eps, a->dW3};
    if (nt)
        CHK(pngpd_bwd_e_impl(x, B, N, T, a->w1, a->b1, s1c, t1c, nullptr, is1, nm1, is2, nm2, w.evec, w.evec + 128,
                             w.evec + 256, nullptr, s.w2tx, nt, s.z2t, w.g2t, S, w.pc, w.pR, w.pW2, &WT, stream));
    else
        CHK(pngpd_bwd_e_impl(x, B, N, T, a->w1, a->b1, s1c, t1c, s.w2p, is1, nm1, is2, nm2, w.evec, w.evec + 128,
                             w.evec + 256, s.w2tp, nullptr, 0, s.z2t, w.g2t, S, w.pc, w.pR, w.pW2, &WT, stream));
    {
        RFArgs A; A.M = (double)B * N; A.eps = (double)a->eps; A.momentum = 0.0;
        RFSeg g{}; g.in = w.pW2; g.out = a->dW2; g.outer = 1; g.R = d.blk; g.n = 128 * 64; g.kind = RF_F32;
        RFSeg c{}; c.in = w.pc; c.out = w.c12; c.outer = 1; c.R = d.blk; c.n = 128; c.kind = RF_F64;
        RFSeg r{}; r.in = w.pR; r.out = w.Rb; r.outer = B; r.R = S; r.n = 192; r.kind = RF_F64;
        RFSeg z{}; z.kind = RF_ZERO; z.f0 = a->db1; z.f1 = a->db2; z.f2 = a->db3;
        A.seg[0] = g; A.seg[1] = c; A.seg[2] = r; A.seg[3] = z;
        CHK(pngpd_reduce_fin_launch(A, 4, stream));
    }
    return pngpd_dw1_finalize(w.Rb, T, s.mom, B, N, w.c12, s.stats1, a->w1, a->b1, a->g1, a->eps, a->dW1, a->dg1,
                              a->dbe1, a->dT, stream);
}

// ---------------------------------------------------------------------------------------
// FC head: fc1 -> BatchNorm1d(batch stats) -> ReLU -> fc2 -> BN -> ReLU -> fc3 -> tail
// ---------------------------------------------------------------------------------------
namespace {
struct HeadSave { float *z1, *y1, *mean1, *var1, *z2, *y2, *mean2, *var2; };
bool carve_head_save(Carve &c, const pngpd_head_train_t *a, HeadSave &s) {
    const size_t B = a->B;
    s.z1 = c.take<float>(B * a->H1); s.y1 = c.take<float>(B * a->H1);
    s.mean1 = c.take<float>(a->H1); s.var1 = c.take<float>(a->H1);
    s.z2 = c.take<float>(B * a->H2); s.y2 = c.take<float>(B * a->H2);
    s.mean2 = c.take<float>(a->H2); s.var2 = c.take<float>(a->H2);
    return c.ok;
}
struct HeadScratch { float *dl, *dy2, *dz2, *dy1, *dz1; };
bool carve_head_scratch(Carve &c, const pngpd_head_train_t *a, HeadScratch &w) {
    const size_t B = a->B;
    w.dl = c.take<float>(B * a->k);
    w.dy2 = c.take<float>(B * a->H2); w.dz2 = c.take<float>(B * a->H2);
    w.dy1 = c.take<float>(B * a->H1); w.dz1 = c.take<float>(B * a->H1);
    return c.ok;
}
bool head_ok(const pngpd_head_train_t *a) {
    return a && a->B > 0 && a->K0 > 0 && a->H1 > 0 && a->H2 > 0 && a->k > 0 && a->inp && a->W1 && a->b1 && a->g1 &&
           a->be1 && a->W2 && a->b2 && a->g2 && a->be2 && a->W3 && a->b3 && a->out && a->save &&
           (a->epilogue == PNGPD_EPI_NONE || a->epilogue == PNGPD_EPI_ADD_IDEN3 || a->epilogue == PNGPD_EPI_LOG_SOFTMAX);
}
}  // namespace

size_t pngpd_head_train_save_bytes(const pngpd_head_train_t *a) {
    if (!a || a->B <= 0 || a->H1 <= 0 || a->H2 <= 0) return 0;
    Carve c(nullptr, 0); HeadSave s;
    carve_head_save(c, a, s);
    return c.off;
}

size_t pngpd_head_train_scratch_bytes(const pngpd_head_train_t *a) {
    if (!a || a->B <= 0 || a->H1 <= 0 || a->H2 <= 0 || a->k <= 0) return 0;
    Carve c(nullptr, 0); HeadScratch w;
    carve_head_scratch(c, a, w);
    return c.off;
}

int pngpd_head_train_fwd(const pngpd_head_train_t *a, void *stream) {
    if (!head_ok(a)) return PNGPD_ERR_INVALID_ARG;
    Carve c(a->save, a->save_bytes); HeadSave s;
    if (!carve_head_save(c, a, s)) return PNGPD_ERR_WORKSPACE;
    CHK(pngpd_fc_fwd(a->inp, a->B, a->K0, a->W1, a->b1, a->H1, PNGPD_EPI_NONE, s.z1, stream));
    CHK(pngpd_bn1d_fwd_train(s.z1, a->B, a->H1, a->g1, a->be1, a->eps, 1, s.y1, s.mean1, s.var1, a->momentum, a->rm1,
                             a->rv1, a->nbt1, stream));
    CHK(pngpd_fc_fwd(s.y1, a->B, a->H1, a->W2, a->b2, a->H2, PNGPD_EPI_NONE, s.z2, stream));
    CHK(pngpd_bn1d_fwd_train(s.z2, a->B, a->H2, a->g2, a->be2, a->eps, 1, s.y2, s.mean2, s.var2, a->momentum, a->rm2,
                             a->rv2, a->nbt2, stream));
    CHK(pngpd_fc_fwd(s.y2, a->B, a->H2, a->W3, a->b3, a->k, a->epilogue, a->out, stream));
    if (a->target) {   // F.nll_loss(output, target), main_1v.py:74
        if (a->epilogue != PNGPD_EPI_LOG_SOFTMAX || !a->loss) return PNGPD_ERR_INVALID_ARG;
        return pngpd_nll_fwd(a->out, a->target, a->B, a->k, a->loss_mean, a->loss, stream);
    }
    return PNGPD_OK;
}

int pngpd_head_train_bwd(const pngpd_head_train_t *a, void *stream) {
    if (!head_ok(a) || (!a->gout && !(a->target && a->gloss)) || !a->scratch || !a->dW1 || !a->db1 || !a->dg1 || !a->dbe1 || !a->dW2 || !a->db2 ||
        !a->dg2 || !a->dbe2 || !a->dW3 || !a->db3)
        return PNGPD_ERR_INVALID_ARG;
    Carve c(a->save, a->save_bytes), cw(a->scratch, a->scratch_bytes);
    HeadSave s; HeadScratch w;
    if (!carve_head_save(c, a, s) || !carve_head_scratch(cw, a, w)) return PNGPD_ERR_WORKSPACE;
    const float *g = a->gout;
    if (a->target && a->gloss) {
        if (a->epilogue != PNGPD_EPI_LOG_SOFTMAX) return PNGPD_ERR_INVALID_ARG;
        CHK(pngpd_nll_log_softmax_bwd(a->gout, a->gloss, a->target, a->out, a->B, a->k, a->loss_mean, w.dl, stream));
        g = w.dl;
    } else if (a->epilogue == PNGPD_EPI_LOG_SOFTMAX) {
        CHK(pngpd_log_softmax_bwd(a->gout, a->out, a->B, a->k, w.dl, stream));
        g = w.dl;
    }
    CHK(pngpd_fc_bwd(g, s.y2, a->W3, a->B, a->H2, a->k, a->dW3, w.dy2, a->db3, stream));
    CHK(pngpd_bn1d_bwd(w.dy2, s.z2, s.y2, a->B, a->H2, a->g2, s.mean2, s.var2, a->eps, 1, w.dz2, a->dg2, a->dbe2,
                       stream));
    // fc2.bias / fc1.bias sit ahead of a train-mode BatchNorm: their gradient is exactly zero (as for the conv biases)
    CHK(pngpd_fc_bwd_impl(w.dz2, s.y1, a->W2, a->B, a->H1, a->H2, a->dW2, w.dy1, a->db2, 1, stream));
    CHK(pngpd_bn1d_bwd(w.dy1, s.z1, s.y1, a->B, a->H1, a->g1, s.mean1, s.var1, a->eps, 1, w.dz1, a->dg1, a->dbe1,
                       stream));
    return pngpd_fc_bwd_impl(w.dz1, a->inp, a->W1, a->B, a->K0, a->H1, a->dW1, a->dinp, a->db1, 1, stream);
}

}  // extern "C"
