// Library-internal launch helpers shared by pngpd_train_glue.hip (kernels) and pngpd_train_step.hip (the fused
// per-direction training entries).  Not part of the C ABI.
#pragma once
#include "pngpd_common.h"

enum { RF_F64 = 0, RF_F32 = 1, RF_BN2 = 2, RF_BN3 = 3, RF_EPREP = 4, RF_ZERO = 5 };

// One segment of reduce_fin_kernel: in (outer, R, n[, planes]) fp32 partials; operands of the finalize kinds:
//   RF_BN2:   p0 = b2, p1 = g2, p2 = be2, rm / rv / nbt, f0 = chan2 (4,128), s0 = stats2 f64[256]
//   RF_BN3:   p0 = b3, p1 = g3, rm / rv / nbt, s0 = stats3 f64[2048]
//   RF_EPREP: p0 = g2, d0 = stats2, f0 = dg2, f1 = dbe2, f2 = evec (3,128)
//   RF_ZERO:  f0 (64), f1 (128), f2 (1024) zero-filled (any may be NULL)
struct RFSeg {
    const float *in; void *out; int outer, R, n, bpo, kind, vec;
    const float *p0, *p1, *p2; const double *d0;
    float *rm, *rv; long long *nbt;
    float *f0, *f1, *f2; double *s0;
};
struct RFArgs { RFSeg seg[4]; int first[5]; double M, eps, momentum; };
int pngpd_reduce_fin_launch(RFArgs &A, int nseg, void *stream);

#define PACK_MAX_JOBS 6
struct PackJob { const float *W; const float *sgn_src; void *out; int C, K, transpose, src_packed, fmt; };
// mom_x != NULL: mom_B more workgroups behind the pack jobs compute the per-cloud input moments (pass A) of x (B,3,N)
// in the same launch — the two are independent and both precede bn1's finalize in the fused forward.
struct PackArgs { PackJob job[PACK_MAX_JOBS]; int first[PACK_MAX_JOBS + 1]; const float *mom_x; double *mom; int mom_N, mom_B; };
int pngpd_train_pack_launch(PackArgs &A, int njobs, void *stream);

// The VALU order of pngpd_trunk_pool_refine that reproduces v_mfma_f32_32x32x2_f32 bit for bit (probed on the device:
// tests/test_gpu_refine.py::test_valu_variant_matches_matrix_pipe).
#define PNGPD_REFINE_VALU_VARIANT 1

// pngpd_fc_bwd with the bias gradient written as an exact zero (zero_db != 0): layers that feed a train-mode BatchNorm.
int pngpd_fc_bwd_impl(const float *g, const float *x, const float *W, int B, int K, int Nout,
                      float *dW, float *dx, float *db, int zero_db, void *stream);

// Passes of the trunk backward with an optional TAIL: independent finalize work carried as extra workgroups of the same
// launch (pngpd_glue_bodies.h).  tail == NULL: exactly the public entries.
struct DW3Args;
struct ACvecArgs;
int pngpd_bwd_gather_impl(const float *x, int B, int N, const float *trans, const float *w1, const float *b1,
                          const float *s1c, const float *t1c, const float *w2p, const void *w2x, int nterms,
                          const float *s2c, const float *t2c, const int *idx, const float *coef,
                          int clouds_per_range, float *Gp, const ACvecArgs *tail, void *stream);
int pngpd_bwd_e_impl(const float *x, int B, int N, const float *trans, const float *w1, const float *b1,
                     const float *s1c, const float *t1c, const float *w2p, const float *is1, const float *nm1,
                     const float *is2, const float *nm2, const float *a1m, const float *a2m, const float *dsc2,
                     const float *w2tp, const void *w2tx, int nterms, const float *z2t, const float *g2t, int S,
                     float *pc, float *pR, float *pW2, const DW3Args *tail, void *stream);
