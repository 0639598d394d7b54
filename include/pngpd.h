/*
 * pngpd.h — C ABI of libpngpd.so, the MI355X (gfx950) implementation of the
 * PointNetGPD grasp-evaluation hot path.
 *
 * The reference (lianghongzhuo/PointNetGPD) has no FFI for this path: it is pure
 * Python dispatching ATen ops.  Each entry point below therefore cites the reference
 * *call site* it replaces (paths relative to the reference root).  The boundary is
 * plain C: raw device pointers, sizes, and a HIP stream passed as `void*`
 * (a `hipStream_t`; NULL = the default stream).  No torch types, no allocation,
 * no host/device synchronisation, no retained pointers.  Every function returns
 * PNGPD_OK (0) or a PNGPD_ERR_* code; `pngpd_strerror` names it.
 *
 * Layout conventions (all row-major, contiguous, device memory unless stated):
 *   clouds      x      (B,3,N)  fp32   channel-major, as Dataset.__getitem__ +
 *                                       default_collate produce (dataset.py:440-444)
 *   pooled      g      (B,1024) fp32   output of MaxPool1d+view (pointnet.py:32-33,148-149)
 *   transforms  trans  (B,3,3)  fp32   STN3d output (pointnet.py:44)
 *   weights     W      (Cout,Cin) fp32 Conv1d(k=1).weight[:,:,0] / Linear.weight
 */
#ifndef PNGPD_H
#define PNGPD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNGPD_ABI_VERSION 7

enum {
    PNGPD_OK = 0,
    PNGPD_ERR_INVALID_ARG = 1,   /* NULL pointer, non-positive size, unsupported shape */
    PNGPD_ERR_WORKSPACE = 2,     /* workspace smaller than pngpd_*_workspace_bytes()  */
    PNGPD_ERR_UNSUPPORTED = 3,
    PNGPD_ERR_HIP = 100          /* PNGPD_ERR_HIP + hipError_t of the failed launch   */
};

/* Weight layouts produced by pngpd_fold_conv_bn. */
enum {
    PNGPD_LAYOUT_ROWMAJOR = 0,   /* (C,K) as given                                     */
    PNGPD_LAYOUT_MFMA_B = 1      /* v_mfma_f32_32x32x2_f32 B-fragment order, see DESIGN */
};

/* Epilogues of pngpd_fc_fwd. */
enum {
    PNGPD_EPI_NONE = 0,
    PNGPD_EPI_RELU = 1,          /* F.relu(bn(fc(x))) with BN folded   pointnet.py:35-36,191-192 */
    PNGPD_EPI_ADD_IDEN3 = 2,     /* fc3(x) + eye(3).view(1,9)          pointnet.py:37-43         */
    PNGPD_EPI_LOG_SOFTMAX = 3    /* F.log_softmax(fc3(x), dim=-1)      pointnet.py:193-194       */
};

int pngpd_abi_version(void);
const char *pngpd_strerror(int code);
/*
 * Fold an eval-mode BatchNorm1d into the preceding 1x1 Conv1d / Linear and lay the
 * weight out for the kernels.  Replaces the eval-mode `self.bnX(self.convX(x))`
 * pairs of pointnet.py:29-31,35-36,144-147,191-192:
 *     s = gamma / sqrt(var + eps);  Wf = W * s[:,None];  bf = (b - mean) * s + beta
 * gamma == NULL means "no BatchNorm" (fc3 layers): Wf = W, bf = b.
 * var == NULL (gamma != NULL) means "scale only": s = gamma, bf = b (used to sign-fold W3 and to
 * re-layout raw weights for the training passes).  b == NULL is read as zeros.
 * layout PNGPD_LAYOUT_MFMA_B requires C % 32 == 0 and K % 8 == 0.
 */
int pngpd_fold_conv_bn(const float *W, const float *b, const float *gamma, const float *beta,
                       const float *mean, const float *var, float eps, int C, int K, int layout,
                       float *Wf, float *bf, void *stream);

/*
 * Fold EVERY layer of a model in ONE launch (round 6): the eval-mode forward re-derives its folded / packed weights
 * from the live parameters on every call — what the reference does implicitly by reading `self.convX.weight` /
 * `self.bnX.running_mean` in every forward (pointnet.py:29-31,35-37,144-147,191-193) — instead of trusting a cache
 * keyed by tensor version counters, which an in-place edit through `.data` does not bump.  ~13 MB of traffic for a
 * PointNetCls (1.6 M weights in, folded copies out): a few microseconds, no host synchronisation, graph-capturable.
 * Per layer (the arithmetic of pngpd_fold_conv_bn, fp64): any subset of three outputs —
 *     row   (C,K) fp32 row-major            (PNGPD_LAYOUT_ROWMAJOR)
 *     mfma  (C,K) fp32 MFMA_B fragments     (PNGPD_LAYOUT_MFMA_B; C % 32 == 0, K % 8 == 0)
 *     x3    2*C*K halfwords                 (pngpd_split_pack_bf16 of the folded row-major weight; C % 32 == 0, K % 16 == 0)
 * plus the folded bias bf (C).  gamma == NULL: no BatchNorm (Wf = W, bf = b).
 */
#define PNGPD_FOLD_MAX_LAYERS 16
typedef struct pngpd_fold_layer {
    const float *W, *b, *gamma, *beta, *mean, *var;
    float eps;
    int C, K;
    float *row, *mfma;
    void *x3;
    float *bf;
} pngpd_fold_layer_t;
typedef struct pngpd_fold_model {
    int n;                                          /* layers used, 1..PNGPD_FOLD_MAX_LAYERS */
    pngpd_fold_layer_t layer[PNGPD_FOLD_MAX_LAYERS];
} pngpd_fold_model_t;
int pngpd_fold_model(const pngpd_fold_model_t *m, void *stream);

/*
 * Fused per-point MLP 3->64->128->1024 (BN folded, ReLU after layers 1,2 and — iff
 * relu_last — after layer 3) + global max over the N points of each cloud.  The
 * (B,1024,N) activation never leaves the chip.  Replaces, in eval mode,
 *   STN3d trunk        pointnet.py:29-33   (relu_last = 1, trans = NULL)
 *   PointNetfeat trunk pointnet.py:140-149 (relu_last = 0, trans = STN output; the
 *                      transpose/bmm/transpose of :140-143 is applied per point)
 *   x      (B,3,N) fp32;  trans (B,3,3) fp32 or NULL
 *   w1 (64,3) rowmajor folded, b1 (64);  w2p (128,64) MFMA_B folded, b2 (128);
 *   w3p (1024,128) MFMA_B folded, b3 (1024)
 *   out_pool (B,1024) fp32
 *   splits: workgroups per cloud (each takes a range of 64-point tiles; the partial maxima are combined by a
 *           second tiny launch); <= 0 selects pngpd_trunk_infer_splits(B, N, 0).  An explicit argument — the
 *           library holds no process-global tuning state.
 *   workspace: pngpd_trunk_workspace_bytes(B, N, splits) bytes of device scratch (0 when one workgroup per cloud).
 */
int pngpd_trunk_infer_splits(int B, int N, int target_blocks);   /* target_blocks <= 0: the default (1024) */
size_t pngpd_trunk_workspace_bytes(int B, int N, int splits);
int pngpd_trunk_fwd_infer(const float *x, int B, int N, const float *trans,
                          const float *w1, const float *b1, const float *w2p, const float *b2,
                          const float *w3p, const float *b3, int relu_last, int splits,
                          float *out_pool, void *workspace, size_t workspace_bytes, void *stream);

/*
 * out = epilogue(in @ W^T + bias)   in (B,K), W (Nout,K) rowmajor (BN pre-folded), out (B,Nout).
 * Replaces the Linear(+BatchNorm1d eval)(+ReLU) stacks pointnet.py:35-37,191-193 and the
 * `+ iden` (:39-43) / `F.log_softmax` (:194) tails.  K % 4 == 0 (rows are read as 16-byte vectors).  ADD_IDEN3 needs Nout == 9,
 * LOG_SOFTMAX needs Nout <= 32.
 */
int pngpd_fc_fwd(const float *in, int B, int K, const float *W, const float *bias, int Nout,
                 int epilogue, float *out, void *stream);

/*
 * OPT-IN reduced-precision trunks on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate): same
 * contract as pngpd_trunk_fwd_infer, the two GEMM layers evaluated with `nterms` bf16 products per fp32 product:
 *   nterms = 3 ("bf16x3"): a*w ~= a_hi*w_hi + a_hi*w_lo + a_lo*w_hi — relative product error <= ~2^-16; log-probs
 *                          stay within 1e-4 of the fp32 path (inside the 1e-3 contract), not bit-identical;
 *   nterms = 1 ("bf16"):   a_hi*w_hi only — plain bf16 operands (BASELINE configs[2]); error ~2^-8 per product,
 *                          does NOT meet 1e-3 on log-probs in general (measured bounds in tests/test_gpu_bf16.py).
 * x: (B,3,N) clouds, fp32 or — x_is_bf16 — bf16 storage (6 B/point).  Layer 1 (K = 3) stays fp32 on the VALU.
 * w2x / w3x = pngpd_split_pack_bf16 of the BN-folded ROWMAJOR (128,64) / (1024,128) weights (2*C*K halfwords each;
 * nterms = 1 reads only the hi halves).  splits over 128-point tiles (pngpd_trunk_infer_bf_splits; default target
 * 1024 workgroups); workspace >= B*S*1024*4 bytes when S > 1.
 */
int pngpd_split_pack_bf16(const float *W, int C, int K, void *out, void *stream);
int pngpd_trunk_infer_bf_splits(int B, int N, int target_blocks);
int pngpd_trunk_fwd_infer_bf(const void *x, int x_is_bf16, int B, int N, const float *trans,
                             const float *w1, const float *b1, const void *w2x, const float *b2,
                             const void *w3x, const float *b3, int relu_last, int nterms, int splits,
                             float *out_pool, void *workspace, size_t workspace_bytes, void *stream);
/* The same launch, also returning out_arg (B,1024) int32 = the point whose reduced-precision value was the maximum of
 * every (cloud, channel) — the input of pngpd_trunk_pool_refine for EVAL-mode refinement (the folded inference weights
 * take the place of the train-mode affine forms there: s1c = t1c = NULL, s2c = ones, t2c = the folded conv2 bias,
 * g3 = ones, w3 = the folded conv3 weight, row-major).  workspace (splits > 1): B*S*1024 * (4 + 4) bytes.           */
int pngpd_trunk_fwd_infer_bf_arg(const void *x, int x_is_bf16, int B, int N, const float *trans,
                                 const float *w1, const float *b1, const void *w2x, const float *b2,
                                 const void *w3x, const float *b3, int relu_last, int nterms, int splits,
                                 float *out_pool, int *out_arg, void *workspace, size_t workspace_bytes,
                                 void *stream);
/* the same arithmetic for pass C of the training path (pngpd_trunk_fwd_train below): identical outputs/semantics.
 * w2x = split_pack_bf16(raw W2), w3sx = split_pack_bf16(sign(gamma3)*W3).  S = workgroups per cloud
 * (1 <= S <= ceil(N/128)); pmax/parg (B*S,1024), psum (B*S,2,1024), psh (B*S*2,128).  BatchNorm statistics and every
 * accumulator stay fp32/fp64.  z2t: the z2 tiles pngpd_trunk_bn2_stats_bf stored WITH THE SAME nterms (fp32 tiles
 * for 3, bf16 tiles for 1) — read back instead of recomputing layers 1-2 — or NULL.                             */
int pngpd_trunk_fwd_train_bf(const float *x, int B, int N, const float *trans,
                             const float *w1, const float *b1, const float *s1c, const float *t1c,
                             const void *w2x, const float *s2c, const float *t2c, const void *w3sx, int nterms, int S,
                             float *pmax, int *parg, float *psum, float *psh, const void *z2t, void *stream);

/* =======================================================================================
 * Training path (batch-statistics BatchNorm, backward).  The trunk's forward/backward is a
 * sequence of recompute passes (DESIGN.md "Training passes"); the host composes them
 * (pointnetgpd_amd/train.py).  Together they replace the autograd graph that
 * PointNetGPD/main_1v.py:72-76 builds over pointnet.py:29-33 / :140-149.
 * Common arguments: x (B,3,N); trans (B,3,3) or NULL; layer-1/2 per-channel forms
 *   h1 = relu((W1 x' + b1)*s1c + t1c),  h2 = relu((W2 h1)*s2c + t2c)
 * w1 (64,3) raw rowmajor, b1/s1c/t1c (64); w2p (128,64) raw MFMA_B packed, s2c/t2c (128).
 * "blk" below = B * S workgroups: S (workgroups per cloud, 1 <= S <= ceil(N/64)) is an explicit argument of
 * every pass — there is no process-global tuning state; pngpd_trunk_splits() only suggests a value, and
 * the caller sizes the partial buffers with the S it passes.  Partial buffers are reduced by
 * pngpd_reduce_partials* (deterministic, no atomics).
 * ======================================================================================= */
/* S that gives about target_blocks workgroups (<= 0: the default, 1024 = two resident rounds of 2 per CU). */
int pngpd_trunk_splits(int B, int N, int target_blocks);

/* pass A: per-cloud fp64 moments  mom (B,9) = {sx,sy,sz,sxx,sxy,sxz,syy,syz,szz}  (BN1 stats in closed form) */
int pngpd_cloud_moments(const float *x, int B, int N, double *mom, void *stream);

/* pass B: BN2 statistics.  part (blk,128,2) = per-workgroup sum / sum-of-squares of z2 = W2 h1.
 *   z2t (optional, pngpd_trunk_g2t_bytes(B,N) bytes): z2 itself, stored in the lane-major tile layout of the trunk
 *   passes; passes C, D and E given this buffer read it back instead of recomputing layers 1-2 (512 B per point of
 *   HBM traffic each, overlapped, against 6 % / 20 % / 33 % of their matrix work).  NULL: not stored.          */
int pngpd_trunk_bn2_stats(const float *x, int B, int N, const float *trans,
                          const float *w1, const float *b1, const float *s1c, const float *t1c,
                          const float *w2p, int S, float *part, float *z2t, void *stream);

/* pass C: layer 3 with sign-folded weights w3sp = MFMA_B(sign(gamma3) * W3):
 *   pmax/parg (blk,1024): max / argmax over the workgroup's points of z3s = w3s . h2
 *   psum (blk,2,1024):    sum / sum of squares of z3s over valid points
 *   psh  (blk,128):       sum of h2 over valid points (its mean enters pass D's cvec and the closed-form dW3)
 *   z2t  (optional): pass B's stored z2 — read back instead of recomputing layers 1-2 (NULL: recompute from x)   */
int pngpd_trunk_fwd_train(const float *x, int B, int N, const float *trans,
                          const float *w1, const float *b1, const float *s1c, const float *t1c,
                          const float *w2p, const float *s2c, const float *t2c, const float *w3sp, int S,
                          float *pmax, int *parg, float *psum, float *psh, const float *z2t, void *stream);

/* Pool refinement of the reduced-precision modes (precision 1 / 3): zex (B,1024) = the EXACT fp32 value of
 *   z3s[b][c] = (sign(gamma3) W3)[c] . h2[b][:, idx[b][c]]
 * at the arg-max point idx the bf16 / bf16x3 pass C chose (layers 1-2 re-evaluated in fp32 for the B*1024 points, then
 * one 128-long contraction each — the arithmetic of pngpd_trunk_fwd_train, operation for operation, so that wherever
 * idx is the fp32 arg-max the value is bit-identical to the fp32 pass's maximum).  Feed zex to pngpd_pool_finalize
 * (S = 1) in place of the bf16 pass's pmax: the reduced-precision matrix pass then contributes only the CHOICE of the
 * point to max over N of bn3(conv3(.)) (PointNetGPD/model/pointnet.py:31-32, :147-148).
 *   s1c / t1c: layer-1 affine form or both NULL (none).  w2p: fp32 MFMA_B-packed conv2 weight; w3sp: sign-folded MFMA_B-packed conv3 weight (variant 0) or NULL;
 *   w3 / g3: raw conv3 weight (1024,128) and bn3.weight (variants 1-3) or NULL;
 *   variant: 0 = contraction on the fp32 matrix pipe (default), 1-3 = on the VALU (candidate orders of the matrix
 *   instruction's two products; tests/test_gpu_refine.py probes which one reproduces it bit for bit).               */
int pngpd_trunk_pool_refine(const float *x, int B, int N, const float *trans,
                            const float *w1, const float *b1, const float *s1c, const float *t1c,
                            const float *w2p, const float *s2c, const float *t2c,
                            const float *w3sp, const float *w3, const float *g3, const int *idx,
                            int clouds_per_range, int variant, float *zex, void *stream);

/* sparse (arg-extremum) term of dW3:  Gp (ceil(B/clouds_per_range),1024,128),
 *   Gp[r][c][:] = sum_{b in range r} coef[b][c] * h2[b][:, idx[b][c]]                         */
int pngpd_trunk_bwd_gather(const float *x, int B, int N, const float *trans,
                           const float *w1, const float *b1, const float *s1c, const float *t1c,
                           const float *w2p, const float *s2c, const float *t2c,
                           const int *idx, const float *coef, int clouds_per_range, float *Gp, void *stream);

/* backward pass D: g2 = dL/d(bn2 out) per point, handed to pass E in g2t (pngpd_trunk_g2t_bytes(B,N) bytes; an
 *   opaque lane-major tile layout shared by the two kernels: [(b*T+tile)][8][256] float4, value 4*q+e of thread t
 *   = point block (4q+e)>>4, MFMA register (4q+e)&15, channel 32*(t>>6) + (t&31));
 *   pa (blk,128,2) = sum g2, sum g2*zhat2;  ps2 (blk,12,16,64) = raw accumulators of 10 of the 16 32x32 blocks of
 *   sum_points h2 h2^T (slot 3w+q of wave w: blocks (w,w), (w,(w+1)%4), (w,w+2 | w<2); the rest by symmetry).
 *   zhat2 = z2*is2 + nm2;  Ap = MFMA_B(A), A (128,128) symmetric;
 *   dh2 = cvec - h2 A + sum_{c: idx[b][c]==n} coef[b][c] W3[c]; w3 (1024,128) raw row-major.
 *   z2t: pass B's stored z2 (or NULL: layers 1-2 are recomputed from x).                      */
size_t pngpd_trunk_g2t_bytes(int B, int N);
int pngpd_trunk_bwd_d(const float *x, int B, int N, const float *trans,
                      const float *w1, const float *b1, const float *s1c, const float *t1c,
                      const float *w2p, const float *s2c, const float *t2c,
                      const float *is2, const float *nm2, const float *Ap, const float *cvec,
                      const float *w3, const int *idx, const float *coef, const float *z2t, int S,
                      float *g2t, float *pa, float *ps2, void *stream);

/* backward pass E: dz2 = dsc2*(g2 - a1m - zhat2*a2m); dh1 = W2^T dz2 (w2tp = MFMA_B(W2^T as (64,128)));
 *   g1 = dh1*(h1>0); pc (blk,64,2) = sum g1, sum g1*zhat1; pR (blk,64,3) = sum_points g1 x^T (original x);
 *   pW2 (blk,128,64) = sum_points dz2 h1^T — the workgroup's share of dL/dW2.  S as passed to pass D.       */
int pngpd_trunk_bwd_e(const float *x, int B, int N, const float *trans,
                      const float *w1, const float *b1, const float *s1c, const float *t1c,
                      const float *w2p, const float *is1, const float *nm1, const float *is2, const float *nm2,
                      const float *a1m, const float *a2m, const float *dsc2, const float *w2tp,
                      const float *z2t, const float *g2t, int S, float *pc, float *pR, float *pW2, void *stream);

/* The same passes with their contractions on the bf16 matrix cores (the opt-in reduced-precision training modes,
 * BASELINE configs[2]; reference call sites as above: pointnet.py:29-33,140-149 under loss.backward(), main_1v.py:75).
 * nterms = 1 plain bf16 operands, 3 = bf16x3 split products; accumulators, BatchNorm statistics, masks and every
 * output stay fp32 and keep the layouts of the fp32 entry points.  Operand matrices are pngpd_split_pack_bf16
 * outputs: w2x of W2 (128,64), Ax of the symmetric matrix A (128,128) of pngpd_a_cvec_finalize (unpacked from its
 * MFMA_B-packed `Ap` to row-major first), w2tx of W2^T as a (64,128) matrix.  Passes D and E always read z2 back (z2t must be non-NULL).
 * Tile storage: with nterms = 3 z2t / g2t are the fp32 tiles of the fp32 entry points (pngpd_trunk_g2t_bytes);
 * with nterms = 1 (plain bf16, BASELINE configs[2] "bf16 storage") they are bf16 tiles of HALF that size —
 * [(b*T + tile)][4][256] x 16 bytes, value v = 8i + e of thread t is half-word e of quad i — written by
 * pngpd_trunk_bn2_stats_bf / pngpd_trunk_bwd_d_bf and read by pngpd_trunk_fwd_train_bf / _bwd_d_bf / _bwd_e_bf.  */
int pngpd_trunk_bn2_stats_bf(const float *x, int B, int N, const float *trans,
                             const float *w1, const float *b1, const float *s1c, const float *t1c,
                             const void *w2x, int nterms, int S, float *part, void *z2t, void *stream);
int pngpd_trunk_bwd_gather_bf(const float *x, int B, int N, const float *trans,
                              const float *w1, const float *b1, const float *s1c, const float *t1c,
                              const void *w2x, int nterms, const float *s2c, const float *t2c,
                              const int *idx, const float *coef, int clouds_per_range, float *Gp, void *stream);
int pngpd_trunk_bwd_d_bf(const float *x, int B, int N, const float *s2c, const float *t2c,
                         const float *is2, const float *nm2, const void *Ax, int nterms, const float *cvec,
                         const float *w3, const int *idx, const float *coef, const void *z2t, int S,
                         void *g2t, float *pa, float *ps2, void *stream);
int pngpd_trunk_bwd_e_bf(const float *x, int B, int N, const float *trans,
                         const float *w1, const float *b1, const float *s1c, const float *t1c,
                         const float *is1, const float *nm1, const float *is2, const float *nm2,
                         const float *a1m, const float *a2m, const float *dsc2, const void *w2tx, int nterms,
                         const void *z2t, const void *g2t, int S, float *pc, float *pR, float *pW2, void *stream);

/* Backward of a Linear layer y = x W^T + b (pointnet.py:35-37,191-193; loss.backward() of main_1v.py:75) in one
 * launch, operands read in place: g (B,Nout) upstream gradient, x (B,K) the layer input, W (Nout,K) ->
 * dW (Nout,K) = g^T x, db (Nout) = sum_b g, dx (B,K) = g W (dx may be NULL).                                */
int pngpd_fc_bwd(const float *g, const float *x, const float *W, int B, int K, int Nout,
                 float *dW, float *dx, float *db, void *stream);

/* BatchNorm1d over the batch dimension for the FC stacks (pointnet.py:35-36,191-192, train mode), optional fused
 * ReLU; biased batch mean/var returned; running_mean / running_var / num_batches_tracked (nullable) updated in
 * place like nn.BatchNorm1d (momentum, unbiased variance). */
int pngpd_bn1d_fwd_train(const float *z, int B, int C, const float *gamma, const float *beta, float eps,
                         int relu, float *y, float *mean, float *var, float momentum, float *rm, float *rv,
                         long long *nbt, void *stream);
int pngpd_bn1d_bwd(const float *dy, const float *z, const float *y, int B, int C, const float *gamma,
                   const float *mean, const float *var, float eps, int relu,
                   float *dz, float *dgamma, float *dbeta, void *stream);
/* backward of F.log_softmax (pointnet.py:194): dlogits = g - exp(logp) * rowsum(g) */
int pngpd_log_softmax_bwd(const float *g, const float *logp, int B, int K, float *dlogits, void *stream);
/* F.nll_loss (main_1v.py:74; test(): :101 with reduction="sum") on log-probabilities, and its backward fused with
 * log_softmax's (SURVEY.md 8b `pngpd_logsoftmax_nll_fwd/bwd`):
 *   fwd: *loss = -sum_b logp[b][target[b]]  (/ B when mean)                      — one workgroup, fp64 accumulation
 *   bwd: G = (g ? g : 0) + onehot(target) * -(*gloss [/ B]);  dlogits = G - exp(logp) * rowsum(G)                   */
int pngpd_nll_fwd(const float *logp, const long long *target, int B, int K, int mean, float *loss, void *stream);
int pngpd_nll_log_softmax_bwd(const float *g, const float *gloss, const long long *target, const float *logp, int B,
                              int K, int mean, float *dlogits, void *stream);

/* ---- finalize kernels: the parameter-sized fp64 algebra between the passes (pngpd_train_glue.hip) ----
 * stats1 f64[140] = mx[3], Cx[9], mu1[64], var1[64];  stats2 f64[256] = mu2r[128], var2[128];
 * stats3 f64[2048] = mu3s[1024], var3[1024] (of the sign-folded z3s);  chan1 f32 (4,64) = s1c,t1c,is1,nm1;
 * chan2 f32 (4,128) = s2c,t2c,is2,nm2.  rm/rv/nbt: BatchNorm running_mean / running_var / num_batches_tracked
 * (nullable), updated in place with `momentum` and the unbiased variance like nn.BatchNorm1d.            */
int pngpd_bn1_finalize(const double *mom, const float *trans, int B, int N, const float *w1, const float *b1,
                       const float *g1, const float *be1, float eps, float momentum, float *rm, float *rv,
                       long long *nbt, float *chan1, double *stats1, void *stream);
/* tot2 f64 (128,2) / tot3 f64 (2,1024): pngpd_reduce_partials of the pass-B / pass-C partial sums */
int pngpd_bn2_finalize(const double *tot2, int B, int N, const float *b2, const float *g2,
                       const float *be2, float eps, float momentum, float *rm, float *rv, long long *nbt,
                       float *chan2, double *stats2, void *stream);
int pngpd_bn3_finalize(const double *tot3, int B, int N, const float *b3, const float *g3,
                       float momentum, float *rm, float *rv, long long *nbt, double *stats3, void *stream);
/* pooled (B,1024) = [relu](g3*zhat + be3), idx (B,1024) arg-extremum point, zhat (B,1024).  parg == idx (in place) is
 * allowed when S == 1 — the pool refinement's second call, which leaves idx unchanged.                              */
int pngpd_pool_finalize(const float *pmax, const int *parg, int B, int S, const double *stats3, const float *g3,
                        const float *be3, float eps, int relu_last, float *pooled, int *idx, float *zhat,
                        void *stream);
/* coef (B,1024) = masked dp * g3/sig3; dg3, dbe3 (1024); m12 f64[2048] = m1, m2 */
int pngpd_bn3_bwd_prep(const float *dp, const float *pooled, const float *zhat, int B, int N, const float *g3,
                       const double *stats3, float eps, int relu_last, float *coef, float *dg3, float *dbe3,
                       double *m12, void *stream);
/* out (outer,n) f64 = sum over r of in (outer,R,n) f32 — deterministic reduction of per-workgroup partials */
int pngpd_reduce_partials(const float *in, int outer, int R, int n, double *out, void *stream);
/* up to four such reductions in one launch (in_i == NULL: slot unused) */
int pngpd_reduce_partials4(const float *in0, int outer0, int R0, int n0, double *out0,
                           const float *in1, int outer1, int R1, int n1, double *out1,
                           const float *in2, int outer2, int R2, int n2, double *out2,
                           const float *in3, int outer3, int R3, int n3, double *out3, void *stream);
/* the operands of pngpd_trunk_bwd_d: Ap = MFMA_B(A) (128x128), cvec (128); sh f64 (128) = sum of h2 (pass C) */
int pngpd_a_cvec_finalize(const double *sh, int B, int N, const float *w3, const float *g3, const double *stats3,
                          const double *m12, float eps, float *Ap, float *cvec, void *stream);
/* dW3 (1024,128) from G f64 (1024,128), S2c f64 (12,16,64) = the reduced ps2 blocks of pass D, sh f64 (128) */
int pngpd_dw3_finalize(const double *G, const double *S2c, const double *sh, int B, int N, const float *w3,
                       const float *g3, const double *stats3, const double *m12, float eps, float *dW3,
                       void *stream);
/* a12 f64 (128,2) = sum g2, sum g2*zhat2 -> dg2, dbe2 (128); evec (3,128) = a1/M, a2/M, g2/sig2: the
 * operands of pngpd_trunk_bwd_e */
int pngpd_bwd_e_prep(const double *a12, int B, int N, const float *g2, const double *stats2, float eps,
                     float *dg2, float *dbe2, float *evec, void *stream);
/* dW1 (64,3), dg1, dbe1 (64); dT (B,3,3) or NULL.  Rb f64 (B,64,3), c12 f64 (64,2) */
int pngpd_dw1_finalize(const double *Rb, const float *trans, const double *mom, int B, int N, const double *c12,
                       const double *stats1, const float *w1, const float *b1, const float *g1, float eps,
                       float *dW1, float *dg1, float *dbe1, float *dT, void *stream);

/* =======================================================================================
 * One entry per direction of the training graph (pngpd_train_step.hip).
 *
 * A training step of the reference (main_1v.py:72-76: forward, nll_loss, loss.backward(), optimizer.step()) walks
 * the trunk of STN3d (pointnet.py:29-33), its FC stack (:35-43), the trunk of PointNetfeat (:140-149) and the head
 * of PointNetCls (:191-194).  Each of those four pieces is ONE call here per direction: the entry enqueues every
 * pass / finalize kernel of the piece on `stream` in the order — and with the arithmetic — of the per-pass entry
 * points above (results are bit-identical to calling those one by one), using caller-provided buffers:
 *   save     what the forward leaves for the backward; must persist (and not be written) between the two calls
 *   scratch  transient; only has to outlive the kernels one call enqueues (a single per-stream buffer will do)
 * sizes: pngpd_*_train_save_bytes / _scratch_bytes of the same descriptor (only the dimension / mode fields are read).
 * No allocation, no synchronisation, no retained pointers.  Gradient outputs are written, not accumulated.
 * ======================================================================================= */
typedef struct pngpd_trunk_train {
    const float *x;        /* (B,3,N) clouds                                                                       */
    const float *trans;    /* (B,3,3) per-cloud input transform (PointNetfeat, pointnet.py:140-143) or NULL (STN3d) */
    int B, N;
    int S;                 /* workgroups per cloud, 1 <= S <= ceil(N/64) (pngpd_trunk_splits)                        */
    int relu_last;         /* ReLU after bn3 (STN3d trunk, pointnet.py:31) or not (PointNetfeat, :147)               */
    int precision;         /* arithmetic of the contractions: 0 fp32 (exact), 3 bf16x3, 1 plain bf16                 */
    int fp32_side;         /* precision != 0: keep passes B / gather / D / E on the fp32 kernels                     */
    int refine;            /* precision != 0: pooled maxima re-evaluated in exact fp32 at the chosen arg-max points
                              (pngpd_trunk_pool_refine): 0 off, 1 on the fp32 matrix pipe, 2 on the VALU                 */
    int need_bwd;          /* forward: a backward will follow (keep z2 and the transposed layer-2 weights)           */
    float eps, momentum;   /* of the three BatchNorm1d layers                                                        */
    const float *w1, *b1, *g1, *be1;   /* conv1.weight (64,3), conv1.bias, bn1.weight, bn1.bias                      */
    const float *w2, *b2, *g2, *be2;   /* conv2 (128,64) ...                                                         */
    const float *w3, *b3, *g3, *be3;   /* conv3 (1024,128) ...                                                       */
    float *rm1, *rv1; long long *nbt1; /* bnX.running_mean / running_var / num_batches_tracked, updated in place by  */
    float *rm2, *rv2; long long *nbt2; /* the forward like nn.BatchNorm1d (momentum, unbiased variance); all nullable */
    float *rm3, *rv3; long long *nbt3;
    float *pooled;         /* (B,1024) forward output: max over N of bn3(conv3(..)) [ReLU]                           */
    int *idx;              /* (B,1024) forward output: arg-extremum point of every (cloud, channel)                  */
    float *zhat;           /* (B,1024) forward output: normalised pre-affine value at that point                     */
    const float *dp;       /* (B,1024) backward input: dL/dpooled                                                    */
    float *dW1, *db1, *dg1, *dbe1;     /* backward outputs, shaped like the parameters; the conv-bias gradients are  */
    float *dW2, *db2, *dg2, *dbe2;     /* exactly zero ahead of a train-mode BatchNorm and are written as zeros (db*  */
    float *dW3, *db3, *dg3, *dbe3;     /* may be NULL)                                                               */
    float *dT;             /* (B,3,3) dL/dtrans or NULL                                                              */
    void *save;    size_t save_bytes;
    void *scratch; size_t scratch_bytes;
} pngpd_trunk_train_t;

size_t pngpd_trunk_train_save_bytes(const pngpd_trunk_train_t *a);      /* reads B, N, S, precision, fp32_side */
size_t pngpd_trunk_train_scratch_bytes(const pngpd_trunk_train_t *a);
/* forward: x [, trans], parameters -> pooled, idx, zhat; running statistics updated */
int pngpd_trunk_train_fwd(const pngpd_trunk_train_t *a, void *stream);
/* backward: dp + the forward's pooled / idx / zhat / save -> dW*, db*, dg*, dbe* [, dT] */
int pngpd_trunk_train_bwd(const pngpd_trunk_train_t *a, void *stream);

typedef struct pngpd_head_train {
    const float *inp;      /* (B,K0) pooled features                                                                 */
    int B, K0, H1, H2, k;  /* fc1 (H1,K0), fc2 (H2,H1), fc3 (k,H2): 1024 / 512 / 256 / 9 or classes                  */
    int epilogue;          /* PNGPD_EPI_ADD_IDEN3 (pointnet.py:37-43), PNGPD_EPI_LOG_SOFTMAX (:194) or PNGPD_EPI_NONE */
    float eps, momentum;
    const float *W1, *b1, *g1, *be1;   /* fc1.weight, fc1.bias, bn.weight, bn.bias                                   */
    const float *W2, *b2, *g2, *be2;
    const float *W3, *b3;
    float *rm1, *rv1; long long *nbt1;
    float *rm2, *rv2; long long *nbt2;
    float *out;            /* (B,k) forward output (read again by the backward of log_softmax)                       */
    const float *gout;     /* (B,k) backward input: dL/dout                                                          */
    float *dinp;           /* (B,K0) backward output or NULL                                                         */
    float *dW1, *db1, *dg1, *dbe1, *dW2, *db2, *dg2, *dbe2, *dW3, *db3;   /* db1 / db2 (biases ahead of a train-mode
                              BatchNorm: exactly zero in exact arithmetic) are written as exact zeros               */
    void *save;    size_t save_bytes;
    void *scratch; size_t scratch_bytes;   /* backward only */
    /* F.nll_loss(output, target) of main_1v.py:74 inside the same two calls (PNGPD_EPI_LOG_SOFTMAX only; all NULL / 0
     * = no loss).  forward: *loss = -sum_b out[b][target[b]] (/ B when loss_mean).  backward: the upstream of `out` is
     * gout (may be NULL when only the loss is differentiated) PLUS the loss path -(*gloss [/ B]) at [b][target[b]]
     * — F.nll_loss's backward — so dlogits = gloss' (softmax - onehot) in one launch and no ATen kernel is left in a
     * training step.  target values outside [0,k) contribute nothing (F.nll_loss asserts there).                      */
    const long long *target;   /* (B) int64 class indices                                                            */
    float *loss;               /* () fp32                                                                            */
    const float *gloss;        /* () fp32 device scalar dL/dloss (backward)                                          */
    int loss_mean;             /* 1: reduction="mean" (main_1v.py:74), 0: "sum" (the per-rank sum of the DDP step)   */
} pngpd_head_train_t;

/* Diagnostic, used by bench.py's roofline block: a stream of independent matrix instructions and nothing else on every
 * CU (dtype 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_bf16; waves_per_simd 1 or 2; `iters` x 8 instructions per
 * wave).  Time the launch on `stream`; *flops_out (may be NULL) receives the FLOPs it executes, so FLOPs / time is the
 * matrix rate this device SUSTAINS — the ceiling the trunk kernels are priced against next to the nominal peak.
 * sink: >= 512 x CU-count floats of scratch (never written).                                                          */
int pngpd_probe_mfma_rate(int dtype, int waves_per_simd, int iters, float *sink, long long *flops_out, void *stream);

size_t pngpd_struct_bytes(int which);   /* sizeof of pngpd_trunk_train_t (0) / pngpd_head_train_t (1) / pngpd_fold_model_t (2): FFI self-check */
size_t pngpd_head_train_save_bytes(const pngpd_head_train_t *a);        /* reads B, H1, H2 */
size_t pngpd_head_train_scratch_bytes(const pngpd_head_train_t *a);     /* reads B, H1, H2, k */
int pngpd_head_train_fwd(const pngpd_head_train_t *a, void *stream);
int pngpd_head_train_bwd(const pngpd_head_train_t *a, void *stream);

/* Adam (main_1v.py:61 `optim.Adam(model.parameters(), lr=args.lr)`; torch defaults otherwise) over ONE flat
 * parameter / gradient / first-moment / second-moment buffer of n fp32 values (16-byte aligned), one launch:
 *   g' = gscale*g [/ max(*gdiv_dev, 1): a device-resident divisor, e.g. the all-reduced sample count];
 *   m += (g'-m)(1-beta1);  v = beta2 v + (1-beta2) g'^2;
 *   p -= lr/(1-beta1^t) * m / (sqrt(v)/sqrt(1-beta2^t) + eps)
 * t = `step` (>= 1, the count of THIS update) unless step_dev != NULL, lr likewise overridden by lr_dev: device-resident
 * scalars (fp32) for captured graphs; pngpd_adam_step_inc advances step_dev by one.                                  */
int pngpd_adam_flat(float *p, const float *g, float *m, float *v, long long n, float lr, const float *lr_dev,
                    float beta1, float beta2, float eps, float step, const float *step_dev, float gscale,
                    const float *gdiv_dev, void *stream);
int pngpd_adam_step_inc(float *step_dev, void *stream);

/* =======================================================================================
 * Batched in-gripper crop + resample (upstream of the scorer).
 * A grasp frame is 18 doubles: origin[3], M[9] (rows approach, binormal, minor), lo[3], hi[3];
 * a cloud point p is kept iff lo < M (p - origin) < hi component-wise (strict, fp64) —
 * dataset.py:53-69 (training box: origin = jaw centre, +-w/4, +-w/2, +-w/4) and
 * kinect2grasp.py:186-229 (inference box: origin = hand bottom, (0,hand_depth), +-w/2, +-w/4).
 * cloud: (P,3) row-major, fp64 if cloud_is_f64 else fp32 (promoted to fp64, kinect2grasp.py:188).
 * ======================================================================================= */
/* counts (G) = number of in-box points; idx (G,max_keep) = their point indices in ascending
 * order (np.where order), truncated to max_keep.                                              */
int pngpd_crop_count_compact(const void *cloud, int cloud_is_f64, int P, const double *frames, int G,
                             int max_keep, int *counts, int *idx, void *stream);
/* Same, against a cloud ARENA (all clouds of a dataset resident in one (P,3) buffer): grasp g sees only the
 * points [ranges[2g], ranges[2g] + ranges[2g+1]) — the cloud file its sample was drawn with, dataset.py:429-433 —
 * and idx holds arena-absolute indices, so pngpd_crop_resample applies unchanged.  ranges (G,2) int32.  */
int pngpd_crop_count_compact_ranges(const void *arena, int cloud_is_f64, int P, const double *frames,
                                    const int *ranges, int G, int max_keep, int *counts, int *idx, void *stream);
/* Same, with an explicit point list per grasp: grasp g sees arena[gather[g][0..Pg)] in that order (duplicates
 * allowed) — the per-sample cloud of the full-view datasets, Pg rows drawn with replacement from the stack of the
 * randomly chosen view files (dataset.py:252-254).  gather (G,Pg) int32, arena-absolute; idx likewise.        */
int pngpd_crop_count_compact_gather(const void *arena, int cloud_is_f64, int P, const double *frames,
                                    const int *gather, int Pg, int G, int max_keep, int *counts, int *idx,
                                    void *stream);
/* out (G,3,N) fp32 = N resampled in-box points per grasp in the hand frame (the `.T` layout of
 * dataset.py:440-444); valid (G) = count >= min_points (dataset.py:71, kinect2grasp.py:462).
 * mode 0: without replacement iff count > N (dataset.py:439); mode 1: iff count >= N (kinect2grasp.py:474).
 * The draw is uniform over ALL `count` in-box points even when count > max_keep (the index list is then only a
 * prefix): such a grasp re-scans its own cloud, described exactly as for the count pass — cloud/P, and ranges
 * (G,2) or gather (G,Pg) when the count pass used them (else NULL).  sel (G,N) int32 ranks in [0, min(count,
 * max_keep)) injects the draw (tests; list-based, no re-scan); otherwise a device RNG keyed by seed; a
 * without-replacement draw is a uniform random N-subset written in ascending point-index order — the scorer is
 * invariant to the column order.  Invalid grasps are zero-filled.  Dynamic LDS: max(max_keep, N) * 4 bytes.
 * The draw of grasp g is keyed by (seed, g_base + g) and nothing else: g_base = the GLOBAL index of this call's
 * first grasp, so a candidate draws the same points whichever shard (GPU) or scoring batch it is handed to —
 * kinect2grasp.py:454-497 scores every candidate independently of the others.  rows (G) int32 or NULL: destination
 * row of grasp g in `out` (pngpd_batch_keep_rows; < 0 = dropped, nothing written); valid stays indexed by g.   */
int pngpd_crop_resample(const void *cloud, int cloud_is_f64, int P, const double *frames, const int *ranges,
                        const int *gather, int Pg, int G, const int *counts, const int *idx, int max_keep, int N,
                        int mode, int min_points, unsigned long long seed, long long g_base, const int *rows,
                        const int *sel, float *out, unsigned char *valid, void *stream);
/* The inference crop over a spatial index — BASELINE configs[4]: 100,000 candidate hands cropped out of ONE scene
 * (kinect2grasp.py:238-258).  cloud_sorted (P,3) / spheres (C = ceil(P/64), 4) f64: the scene re-ordered along a Morton
 * curve and the bounding sphere of each 64-point chunk (gpg.CloudIndex, as for pngpd_hand_box_counts_indexed); only the
 * chunks whose sphere meets the hand's box are evaluated, with the same fp64 per-point test.  counts as
 * pngpd_crop_count_compact; idx = SORTED positions in ascending order (= the order a plain scan of cloud_sorted visits
 * the in-box points, so pngpd_crop_resample(cloud_sorted, ...) applies unchanged).                                  */
int pngpd_crop_count_compact_indexed(const void *cloud_sorted, int cloud_is_f64, int P, const double *spheres, int C,
                                     const double *frames, int G, int max_keep, int *counts, int *idx, void *stream);
/* pngpd_crop_count_compact_indexed + pngpd_crop_resample in ONE launch: the index list of a hand lives in its
 * workgroup's LDS and never reaches HBM (it was ~1 GB written and read back per 100,000 hands).  Outputs and draw keys
 * exactly as the two calls in sequence.  Dynamic LDS: (max(max_keep, N) + max_keep) * 4 bytes.                      */
int pngpd_crop_indexed(const void *cloud_sorted, int cloud_is_f64, int P, const double *spheres, int C,
                       const double *frames, int G, int max_keep, int N, int mode, int min_points,
                       unsigned long long seed, long long g_base, int *counts, float *out, unsigned char *valid,
                       void *stream);
/* my_collate (main_1v.py:48-50) without a host round trip: sample g of a training batch is kept iff counts[g] >=
 * min_points (dataset.py:71-72) and labels[g] >= 0 (-1 = the reference's label None, dataset.py:446-453).
 * rows (G) int32 = its row in the compacted batch or -1; labels_out[0..n) = the kept labels in order (int64, what
 * F.nll_loss takes, main_1v.py:74); *n_keep (device int32) = n.  One workgroup, any G.                          */
int pngpd_batch_keep_rows(const int *counts, const long long *labels, int G, int min_points, int *rows,
                          long long *labels_out, int *n_keep, void *stream);
/* The per-sample cloud of the full-view datasets, dataset.py:252-254: `pc[np.random.choice(len(pc), size=Pg)]` over
 * the stack of the k_views view files drawn for sample g.  spans (G,k_views,2) int32 = [start, len] of each drawn
 * view in the arena; gather (G,Pg) int32 = Pg uniform rows of the stacked length, arena-absolute, keyed by
 * (seed, g_base + g, draw) — independent of how the epoch is cut into batches.  k_views <= 64.                */
int pngpd_stack_gather_lists(const int *spans, int k_views, int Pg, int G, unsigned long long seed, long long g_base,
                             int *gather, void *stream);
/* One training batch of the HBM-resident data layer in ONE call (what 32 DataLoader workers produce per step in
 * main_1v.py:120-128 through Dataset.__getitem__, dataset.py:420-458 / :244-282, and my_collate, main_1v.py:48-50):
 * [full-view only: pngpd_stack_gather_lists] -> crop count/compact -> pngpd_batch_keep_rows -> pngpd_crop_resample
 * (mode 0) writing straight into the compacted rows.  frames (n_items,18) fp64 and labels (n_items) int64 are
 * per-DATASET tables resident in HBM; item (G) int32 = this batch's rows of them (an epoch's permutation slice).
 * k_views == 0: one-view datasets, spans (G,2) = the arena range of the view drawn for each sample;
 * k_views  > 0: full-view datasets, spans (G,k_views,2), gather_ws (G,Pg) int32 scratch.
 * Scratch: counts (G), idx (G,max_keep), rows (G), valid (G), seg_scratch (4 G) int32 or NULL (with it, the scan of a
 * long full-view sample cloud — Pg >= 16384 — is cut into 4 segments per sample, two launches: same lists).  Outputs: out (>= kept,3,N) fp32, labels_out (G)
 * int64 (first kept entries), *n_keep (int32; device memory, or device-visible pinned host memory — the host then
 * reads the kept count after the stream's next event without a copy launch).  Draws keyed by (seed, g_base + g).   */
int pngpd_train_batch(const void *arena, int arena_is_f64, int P, const double *frames, const long long *labels,
                      const int *item, const int *spans, int k_views, int Pg, int *gather_ws, int G, int max_keep,
                      int N, int min_points, unsigned long long seed, long long g_base, int *counts, int *idx,
                      int *rows, unsigned char *valid, int *seg_scratch, float *out, long long *labels_out, int *n_keep,
                      void *stream);

/* =======================================================================================
 * GPG grasp-candidate sampler, device half (upstream of the crop at inference) —
 * dex-net/src/dexnet/grasping/grasp_sampler.py :: GpgGraspSamplerPcl.sample_grasps :1383-1656.
 * Same cloud convention as the crop entries.  The host half (3x3 eigen-decomposition, pose
 * enumeration, selection logic) is pointnetgpd_amd/gpg.py.
 * ======================================================================================= */
/* :1471-1485  For each of K sample points (queries (K,3) f64): the <= max_nn nearest cloud points with
 * squared distance < radius^2 (ties at the cut -> lower index) and M_out (K,9) f64 = sum over the selected
 * points with non-zero distance of n n^T, n = normals[p] / |normals[p]| (normals (P,3) f64);
 * nsel_out (K) = number of selected points (including a zero-distance one).                     */
int pngpd_gpg_normal_moments(const void *cloud, int cloud_is_f64, const double *normals, int P,
                             const double *queries, int K, double radius, int max_nn, double *M_out,
                             int *nsel_out, void *stream);
/* The same selection over the spatial index of gpg.CloudIndex (Morton-sorted cloud + 64-point chunk spheres, as for
 * pngpd_hand_box_counts_indexed): the chunk spheres bound the max_nn-th nearest distance, only the chunks that can hold
 * a selected point are scanned (a handful on a dense cloud instead of all P points, 11 times).  order (P) int32: sorted
 * position -> ORIGINAL index (ties at the cut go to the lower original index; normals stay in original order).
 * Same selected set as pngpd_gpg_normal_moments and the same order of additions: M is bit-identical (the eigenvector
 * signs of the decomposition that follows can flip on a last-bit change).  max_nn <= 1024, else PNGPD_ERR_UNSUPPORTED.         */
int pngpd_gpg_normal_moments_indexed(const void *cloud_sorted, int cloud_is_f64, const int *order,
                                     const double *normals, int P, const double *spheres, int C,
                                     const double *queries, int K, double radius, int max_nn, double *M_out,
                                     int *nsel_out, void *stream);
/* :336-393 / :405-421  For each of Q hand poses (poses (Q,12) f64 = centre, approach, binormal, minor; unit
 * axes) count the cloud points strictly inside each of num_boxes (1 or 4) boxes of the hand model, boxes
 * (num_boxes,6) f64 = x_lo, x_hi, y_lo, y_hi, z_lo, z_hi in the grasp frame -> counts (Q,num_boxes) int32.
 * check_collision_square's has_p is counts > 0, its points_in_area length is counts.             */
int pngpd_hand_box_counts(const void *cloud, int cloud_is_f64, int P, const double *poses, int Q,
                          const double *boxes, int num_boxes, int *counts, void *stream);
/* Same result (identical counts) through a spatial index, for large clouds / many poses: cloud_sorted is the
 * cloud re-ordered so that consecutive points are close (Morton order), spheres (C,4) f64 = bounding sphere
 * (centre xyz, radius) of every 64-point chunk, C = ceil(P/64).  One wave per pose rejects chunks by their sphere
 * before testing points.                                                                          */
int pngpd_hand_box_counts_indexed(const void *cloud_sorted, int cloud_is_f64, int P, const double *spheres, int C,
                                  const double *poses, int Q, const double *boxes, int num_boxes, int *counts,
                                  void *stream);

/* the same with the number of valid poses on the DEVICE: only poses [0, *valid_units * per_unit) are evaluated (the
 * launch is sized for Q = the buffer's capacity; valid_units == NULL: all Q). */
int pngpd_hand_box_counts_indexed_n(const void *cloud_sorted, int cloud_is_f64, int P, const double *spheres, int C,
                                    const double *poses, int Q, const double *boxes, int num_boxes,
                                    const int *valid_units, int per_unit, int *counts, void *stream);

/* grasp_sampler.py:1486-1506  local frames of K sample points on the device: M (K,3,3) f64 from
 * pngpd_gpg_normal_moments, normals_at (K,3) f64 = all_normal[ind], points (K,3) f64 = the sample points ->
 * frames (K,12) f64 = minor_pc, new_normal (flipped against normals_at, the minor axis with it), major_pc, sample
 * point — the input of pngpd_gpg_enumerate.  `np.linalg.eig(M)` (:1493) is evaluated as numpy evaluates it: LAPACK's
 * DGEEV('N','V') restated for a symmetric 3x3 matrix (csrc/pngpd_gpg_eig3.h: DGEBAL, DGEHD2, DORGHR, DLAHQR + DLANV2,
 * DTREVC3, DGEBAK, 1/DNRM2 — same eigenvalue order, same eigenvector signs; bit-identical to OpenBLAS 0.3.29 on
 * 99.9 % of moment matrices, a few ulp on the rest), because the signs LAPACK returns decide the sweep's enumeration
 * order.  flags (K) int32: 1 = sum(sum(M)) == 0 (:1486 `continue`; the point gets the frame of a hand 1e6 m away, which
 * yields no grasp), 2 = LAPACK would report a complex pair (a double eigenvalue split by rounding; the pair is taken
 * as the double real eigenvalue it is), 4 = no convergence in 30 * 10 QR sweeps, 8 = max|M| outside DGEEV's unscaled
 * range.                                                                                                            */
int pngpd_gpg_frames(const double *M, const double *normals_at, const double *points, int K, double *frames,
                     int *flags, void *stream);

/* ---- selection logic of the sampler on the device (grasp_sampler.py:1524-1650); the 3x3 eigen-decomposition of
 * :1493 runs in pngpd_gpg_frames (or on the host through LAPACK itself: gpg.py eig="lapack").  prm: gripper / sweep constants
 * (layout in pngpd_gpg.hip, built by pointnetgpd_amd/gpg.py).  L live sample points, R rotations, D lateral offsets,
 * S push-in steps; capacity of the per-pose buffers = L*R potential grasps. ---- */
/* :1524-1541  frames (L,12) = minor, normal, major, sample point -> poses (L,R,D,12), ab (L,R,6) = approach, binormal */
int pngpd_gpg_enumerate(const double *frames, int L, int R, int D, const double *prm, double *poses, double *ab,
                        void *stream);
/* :1565-1573  counts (L,R,D,4) of the sweep -> flag/dsel (L*R): potential grasp and its offset; list (L*R) = the
 * potential grasps in (l,r) order, *total their number */
int pngpd_gpg_select(const int *counts, const double *poses, const double *ab, int L, int R, int D, const double *prm,
                     int *flag, int *dsel, int *list, int *total, void *stream);
/* :1531-1573  the lateral sweep AND the selection in one launch, for candidate generation at scale: the D poses of a
 * (sample point, rotation) share one frame and the selection needs only two predicates per pose, so one wave per
 * (l,r) transforms each nearby point once and derives, in closed form, the set of offsets whose opening / collision
 * boxes hold it; every point whose verdict lies within `tol` (in units of one lateral offset; 1e-9 is ample, > 1e20
 * forces the exact path everywhere) of changing is re-evaluated with the exact per-pose arithmetic of
 * pngpd_hand_box_counts, so flag / dsel / list / total equal pngpd_hand_box_counts_indexed + pngpd_gpg_select always.
 * cloud_sorted / spheres: gpg.CloudIndex (as for pngpd_hand_box_counts_indexed); boxes (4,6): opening, left, right,
 * bottom.  masks (L*R,2) uint32 or NULL: bit d of [0] = the opening holds a point at offset d, of [1] = a collision.
 * stats (4) uint64 or NULL, diagnostics, ADDED to: units swept, chunks past the broad phase, chunks evaluated, points
 * sent to the exact path.                                                                                          */
int pngpd_gpg_sweep_select(const void *cloud_sorted, int cloud_is_f64, int P, const double *spheres, int C,
                           const double *poses, const double *ab, int L, int R, int D, const double *boxes,
                           const double *prm, double tol, int *flag, int *dsel, int *list, int *total,
                           unsigned *masks, unsigned long long *stats, void *stream);
/* :1575-1612  push-in poses and their backed-off, table-corrected twins: poses2 (L*R,S,2,12), back / mod (L*R,S,3) */
int pngpd_gpg_pushin(const int *list, const int *total, const int *dsel, const double *poses, const double *ab,
                     const double *frames, int L, int R, int D, int S, const double *prm, double *poses2, double *back,
                     double *mod, void *stream);
/* :1576-1625  the collision / opening tests of ALL push-in poses of a potential grasp in one wave (they share a frame
 * and lie on the approach axis) — straight to found / sfirst, the outputs of pngpd_gpg_finish's first stage, which is
 * then called with counts2 == NULL.  poses2 (L*R,S,2,12) from pngpd_gpg_pushin, *total potential grasps (device), S <= 32
 * (else PNGPD_ERR_UNSUPPORTED); cloud_sorted / spheres / boxes / tol / stats as for pngpd_gpg_sweep_select (tol in
 * push-in steps).  Equal to pngpd_hand_box_counts_indexed_n + the first-accept rule on the exact counts, always.      */
int pngpd_gpg_pushin_sweep(const void *cloud_sorted, int cloud_is_f64, int P, const double *spheres, int C,
                           const double *poses2, const int *total, int L, int R, int S, const double *boxes,
                           int min_open, double tol, int *found, int *sfirst, unsigned long long *stats,
                           void *stream);
/* :1614-1637  counts2 (L*R,S,2,4) [NULL: found / sfirst already hold the first accepted steps] -> first accepted step per potential grasp; res (1 + L + L*R*15) doubles =
 * [n_found, grasps per live sample point (L), rows [bottom, approach, binormal, minor, bottom_modified] in the
 * reference's output order]; found/sfirst/olist (L*R), ototal (1): scratch */
int pngpd_gpg_finish(const int *counts2, const int *list, const int *total, const double *ab, const double *frames,
                     const double *back, const double *mod, int L, int R, int S, int min_open, int *found,
                     int *sfirst, int *olist, int *ototal, double *res, void *stream);

/* =======================================================================================
 * GPD baseline (the comparator model of the paper) and the depth-registration preprocessing — SURVEY.md §8f-4.
 * ======================================================================================= */
/* PointNetGPD/model/dataset.py:88-198 (project_pc / cal_projection) given the in-box points and their normals:
 * points / normals (sum M_g, 3) f64 = the kept points of all G grasps in the hand frame, concatenated in input order;
 * offsets (G+1) int32; widths (G) f64 gripper widths.  out (G,60,60,chann) f64, chann 3 = the normal image of
 * projection order (0,1,2); chann 12 = [occupancy, normal xyz] x orders (0,1,2), (1,2,0), (0,2,1) (:104-116).
 * Points with a NaN normal component are skipped (:97-101).  Bit-identical to numpy (float32 sequential normal sums,
 * last-z-wins pixel assignment).  project_size must be 60 (as in the reference, :221).                       */
int pngpd_gpd_projection(const double *points, const double *normals, const int *offsets, const double *widths,
                         int G, int chann, int project_size, int margin, int voxel_point_num, double *out,
                         void *stream);
/* PointNetGPD/ycb_cloud_generate.py:60-121 registerDepthMap: depth (hd,wd) f64 metres, cam20 = depthK fx,fy,cx,cy |
 * rgbK fx,fy,cx,cy | H_RGBFromDepth rows 0..2 (3x4) -> registered (hr,wr) f64 (zero where nothing lands; the
 * largest transformed depth per pixel, as the reference's `>` keeps).                                        */
int pngpd_depth_register(const double *depth, int hd, int wd, const double *cam20, int hr, int wr,
                         double *registered, void *stream);
/* :124-184 registeredDepthMapToPointCloud (organized=False): the pixels with depth > 0 in row-major order ->
 * xyz (count,3) f64 in the object frame (+ their colours when rgb / rgb_out are given); cam28 = rgbK fx,fy,cx,cy |
 * refFromRGB 3x4 | objFromref 3x4.  xyz / rgb_out must hold h*w rows; *count (device int) receives the number
 * written.  workspace: pngpd_depth_cloud_workspace_bytes(h, w).                                              */
size_t pngpd_depth_cloud_workspace_bytes(int h, int w);
int pngpd_depth_to_cloud(const double *depth, int h, int w, const double *cam28, const unsigned char *rgb,
                         double *xyz, unsigned char *rgb_out, int *count, void *workspace, size_t workspace_bytes,
                         void *stream);
/* PointNetGPD/model/gpd.py:13-24: one convolution stage of GPDClassifier = Conv2d(Cin,Cout,5) (valid) + bias +
 * MaxPool2d(2, stride 2), no activation (as the reference).  in (B,Cin,Hin,Hin), W (Cout,Cin,5,5), out
 * (B,Cout,(Hin-4)/2,(Hin-4)/2) fp32.  The two FC layers of the classifier use pngpd_fc_fwd.                  */
int pngpd_conv5_pool2(const float *in, int B, int Cin, int Hin, const float *W, const float *bias, int Cout,
                      float *out, void *stream);
/* Training form of the stage and its backward (loss.backward() of PointNetGPD/main_1v_gpd.py:105 through gpd.py:13-24).
 * _arg: the same outputs as pngpd_conv5_pool2, plus arg (B,Cout,Hp,Hp) u8 = which window position (2*dy + dx) gave each
 * pooled maximum — the first in row-major order on a tie, as ATen's max_pool2d.
 * _bwd: dout (B,Cout,Hp,Hp) -> dW (Cout,Cin,5,5), db (Cout) and, unless NULL, din (B,Cin,Hin,Hin; needs Hin % 4 == 0 and
 * Hin <= 32: the second stage — the first stage's input is the image).  Deterministic: per-batch-slice partial sums (and the
 * pooled gradients as an (offset, value) list) in `workspace` (pngpd_conv5_pool2_bwd_workspace_bytes), reduced in slice
 * order; no atomics.
 * pngpd_relu_bwd: g <- g * (y > 0) in place, y = the ReLU's OUTPUT (gpd.py:27).                                        */
int pngpd_conv5_pool2_arg(const float *in, int B, int Cin, int Hin, const float *W, const float *bias, int Cout,
                          float *out, unsigned char *arg, void *stream);
size_t pngpd_conv5_pool2_bwd_workspace_bytes(int B, int Cin, int Hin, int Cout);
int pngpd_conv5_pool2_bwd(const float *in, int B, int Cin, int Hin, const float *W, int Cout, const float *dout,
                          const unsigned char *arg, float *dW, float *db, float *din, void *workspace,
                          size_t workspace_bytes, void *stream);
int pngpd_relu_bwd(const float *y, float *g, long long n, void *stream);
/* out = [relu](in @ W^T + bias) for a layer with FEW output tiles and a LONG contraction (gpd.py:27 fc1: 7200 -> 500):
 * K split over workgroups, partial tiles in `workspace` summed in slice order by a second launch (deterministic).
 * K % 8 == 0.  The sum order differs from pngpd_fc_fwd's, so the two agree to rounding, not bit for bit.           */
size_t pngpd_fc_fwd_splitk_workspace_bytes(int B, int K, int Nout);
int pngpd_fc_fwd_splitk(const float *in, int B, int K, const float *W, const float *bias, int Nout, int relu, float *out,
                        void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PNGPD_H */
