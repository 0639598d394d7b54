// libpngpd — an FC stack (fc1 -> BatchNorm1d(batch) -> ReLU -> fc2 -> BN -> ReLU -> fc3 + tail; pointnet.py:35-43 and
// :191-194 under main_1v.py:72-76) in ONE launch per direction.
//
// The work of the launch is the same 32x32 tiles and the same arithmetic as the per-op kernels (pngpd_fc_tile.h), laid
// out as consecutive STAGES of workgroups:
//   forward    [fc1 tiles][fc2 tiles][fc3 tiles]
//   backward   [log-softmax rows][fc3: dx, dW][fc2: dx, dW][fc1: dx, dW]
// Within a stage the workgroups of one 32-channel column draw a ticket when their tile is stored; the LAST one to
// arrive runs that column's BatchNorm (forward: statistics, normalise, ReLU, running statistics; backward: the two
// batch sums and dz) — no workgroup ever waits for a ticket.  A workgroup of the next stage waits, once, for the count
// of finished columns (wg_wait_count): the launcher uses this chained form only when the whole grid is resident on
// the device at once (every workgroup counted on is running or done), and otherwise launches the stages one by one,
// where nothing waits at all.  Hand-off: agent-scope release / acquire around relaxed counters (pngpd_fc_tile.h);
// results do not depend on dispatch order or on the workgroup -> XCD placement.  Column c's tiles sit at workgroup ids
// congruent to c modulo the column count (a multiple of 8 for this model), i.e. on ONE XCD, so the column block the
// BatchNorm re-reads is in that XCD's L2 — a speed choice only.
//
// What it buys: 5 + 6 launches of 5-14 us per stack and step become 1 + 1 (batch <= 128) or 3 + 4 (profiles/r04_*).
#include "pngpd_fc_tile.h"
#include "pngpd_internal.h"

namespace {

enum { TK_EXIT = 0, TK_DONE0 = 1, TK_DONE_A = 2, TK_DONE_B = 3, TK_COLS = 8 };

// One stage = one Linear layer's tiles (+ the BatchNorm that follows it).  The kernels index an array of these with the
// workgroup's (uniform) stage number, so a workgroup loads only its own stage's operands.
struct FwdStage {
    const float *in, *W, *bias, *gamma, *beta;
    float *out, *y, *mean, *var, *rm, *rv; long long *nbt;
    int K, Nout, epi, cols;       // cols: 32-channel column blocks of the layer
    int end;                      // one past the stage's last workgroup id
    int ticket0, done, wait, wait_n;   // counter words: first column ticket, finished-columns count; the count this
                                       // stage waits for (word, target; target 0 = none)
    int bn;
};
struct HeadFwdK {
    FwdStage st[3];
    int B, nrb; float eps, momentum;
    int *tk;
    int wid0, nlaunch, stage_lo, nwords;   // this launch: workgroups [wid0, wid0 + nlaunch), lowest stage stage_lo
};

__global__ __launch_bounds__(256, 4) void head_fwd_fused_kernel(const HeadFwdK a) {
    __shared__ float lds[FC_RED_FLOATS + 4];
    int *flag = (int *)(lds + FC_RED_FLOATS);
    const int wid = (int)blockIdx.x + a.wid0;
    const int s = wid < a.st[0].end ? 0 : wid < a.st[1].end ? 1 : 2;
    const FwdStage &p = a.st[s];
    const int t = wid - (s ? a.st[s - 1].end : 0);
    const int cb = t % p.cols, rb = t / p.cols;
    if (s > a.stage_lo && p.wait_n) wg_wait_count(&a.tk[p.wait], p.wait_n);
    fc_tile<true>(p.in, a.B, p.K, p.W, p.bias, p.Nout, p.epi, p.out, rb, cb, lds);
    if (p.bn && wg_publish_add(&a.tk[p.ticket0 + cb], flag) == a.nrb - 1) {
        wg_acquire();
        if (a.B <= BN_TAIL_NR * BN1D_RL)
            bn1d_fwd_tail<true>(p.out, a.B, p.Nout, cb * 32, p.gamma, p.beta, a.eps, 1, p.y, p.mean, p.var, a.momentum,
                                p.rm, p.rv, p.nbt, lds);
        else
            bn1d_fwd_tail<false>(p.out, a.B, p.Nout, cb * 32, p.gamma, p.beta, a.eps, 1, p.y, p.mean, p.var, a.momentum,
                                 p.rm, p.rv, p.nbt, lds);
        wg_publish_add(&a.tk[p.done], flag);
    }
    wg_exit(a.tk, a.nlaunch, a.nwords);
}

// Backward stage: dx tiles first (workgroups [0, tx) of the stage; their columns feed the BatchNorm backward that the
// next stage waits for), then the dW tiles.
struct BwdStage {
    const float *g, *x, *W;
    float *dW, *dx, *db;
    const float *z, *y, *gamma, *mean, *var;     // BatchNorm behind this layer's INPUT (dx is its dy)
    float *dz, *dgamma, *dbeta;
    int K, Nout, zero_db, tx, end;
    int ticket0, done, wait, wait_n, bn;
};
struct HeadBwdK {
    BwdStage st[3];
    const float *gout, *logp; float *dl;
    int B, k, nrb, n0; float eps;
    int *tk;
    int wid0, nlaunch, stage_lo, nwords;   // stage_lo counts the log-softmax stage as 0, fc3 / fc2 / fc1 as 1 / 2 / 3
};

__global__ __launch_bounds__(256, 4) void head_bwd_fused_kernel(const HeadBwdK a) {
    __shared__ float lds[FC_BWD_LDS_FLOATS + 4];
    int *flag = (int *)(lds + FC_BWD_LDS_FLOATS);
    const int wid = (int)blockIdx.x + a.wid0;
    if (wid < a.n0) {                                       // stage 0: dlogits = g - exp(logp) rowsum(g)
        const int b = wid * 256 + (int)threadIdx.x;
        if (b < a.B) log_softmax_bwd_row(a.gout + (size_t)b * a.k, a.logp + (size_t)b * a.k, a.k, a.dl + (size_t)b * a.k);
        wg_publish_add(&a.tk[TK_DONE0], flag);
    } else {
        const int s = wid < a.st[0].end ? 0 : wid < a.st[1].end ? 1 : 2;
        const BwdStage &p = a.st[s];
        const int t = wid - (s ? a.st[s - 1].end : a.n0);
        if (s + 1 > a.stage_lo && p.wait_n) wg_wait_count(&a.tk[p.wait], p.wait_n);
        const bool is_w = t >= p.tx;
        const int tt = is_w ? t - p.tx : t;
        if ((p.Nout & 7) == 0) fc_bwd_tile<true>(p.g, p.x, p.W, a.B, p.K, p.Nout, is_w, tt, p.dW, p.dx, p.db, p.zero_db, lds);
        else fc_bwd_tile<false>(p.g, p.x, p.W, a.B, p.K, p.Nout, is_w, tt, p.dW, p.dx, p.db, p.zero_db, lds);
        if (!is_w && p.bn) {
            const int col = tt % ((p.K + 31) >> 5);         // dx tile tt = row block * k-blocks + column
            if (wg_publish_add(&a.tk[p.ticket0 + col], flag) == a.nrb - 1) {
                wg_acquire();
                if (a.B <= BN_TAIL_NR * BN1D_RL)
                    bn1d_bwd_tail<true>(p.dx, p.z, p.y, a.B, p.K, col * 32, p.gamma, p.mean, p.var, a.eps, 1, p.dz,
                                        p.dgamma, p.dbeta, lds);
                else
                    bn1d_bwd_tail<false>(p.dx, p.z, p.y, a.B, p.K, col * 32, p.gamma, p.mean, p.var, a.eps, 1, p.dz,
                                         p.dgamma, p.dbeta, lds);
                wg_publish_add(&a.tk[p.done], flag);
            }
        }
    }
    wg_exit(a.tk, a.nlaunch, a.nwords);
}

// Workgroups of `fn` (256 threads, static LDS only) the device holds at once.  A property of the device and the code
// object, read once per device; min(API, 8) per CU (the guide's SGPR rule does not bind below 81 SGPRs).
int resident_capacity(const void *fn, int slot) {
    static int cap[2][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cap[slot][dev] == 0) {
        int per = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, 256, 0) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (per > 8) per = 8;
        cap[slot][dev] = per * cus > 0 ? per * cus : -1;
    }
    return cap[slot][dev] > 0 ? cap[slot][dev] : 0;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

// The shapes the one-launch kernels cover: every layer on the K-split forward tile (what pngpd_fc_fwd picks for them),
// 32-aligned widths.  Everything else stays on the launch-per-op sequence.
bool pngpd_head_fused_ok(const pngpd_head_train_t *a) {
    if (!a->tickets) return false;
    if ((a->K0 & 31) || (a->H1 & 31) || (a->H2 & 31)) return false;
    const long nrb = cdiv(a->B, 32);
    if (nrb * cdiv(a->H1, 32) > 2048 || nrb * cdiv(a->H2, 32) > 2048 || nrb * cdiv(a->k, 32) > 2048) return false;
    return pngpd_head_train_ticket_ints(a) <= 4096;
}

extern "C" size_t pngpd_head_train_ticket_ints(const pngpd_head_train_t *a) {
    if (!a || a->H1 <= 0 || a->H2 <= 0) return 0;
    return (size_t)TK_COLS + cdiv(a->H1, 32) + cdiv(a->H2, 32);
}

int pngpd_head_fwd_fused(const pngpd_head_train_t *a, float *z1, float *y1, float *mean1, float *var1, float *z2,
                         float *y2, float *mean2, float *var2, void *stream) {
    HeadFwdK k{};
    const int nrb = cdiv(a->B, 32), c1 = cdiv(a->H1, 32), c2 = cdiv(a->H2, 32), c3 = cdiv(a->k, 32);
    k.st[0] = FwdStage{a->inp, a->W1, a->b1, a->g1, a->be1, z1, y1, mean1, var1, a->rm1, a->rv1, a->nbt1,
                       a->K0, a->H1, PNGPD_EPI_NONE, c1, nrb * c1, TK_COLS, TK_DONE_A, 0, 0, 1};
    k.st[1] = FwdStage{y1, a->W2, a->b2, a->g2, a->be2, z2, y2, mean2, var2, a->rm2, a->rv2, a->nbt2,
                       a->H1, a->H2, PNGPD_EPI_NONE, c2, nrb * (c1 + c2), TK_COLS + c1, TK_DONE_B, TK_DONE_A, c1, 1};
    k.st[2] = FwdStage{y2, a->W3, a->b3, nullptr, nullptr, a->out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                       a->H2, a->k, a->epilogue, c3, nrb * (c1 + c2 + c3), 0, 0, TK_DONE_B, c2, 0};
    k.B = a->B; k.nrb = nrb; k.eps = a->eps; k.momentum = a->momentum; k.tk = a->tickets;
    k.nwords = (int)pngpd_head_train_ticket_ints(a);
    const int total = k.st[2].end;
    hipStream_t sm = (hipStream_t)stream;
    if (total <= resident_capacity((const void *)head_fwd_fused_kernel, 0)) {
        k.wid0 = 0; k.nlaunch = total; k.stage_lo = 0;
        hipLaunchKernelGGL(head_fwd_fused_kernel, dim3(total), dim3(256), 0, sm, k);
        return pngpd_launch_status();
    }
    for (int s = 0; s < 3; ++s) {
        k.wid0 = s ? k.st[s - 1].end : 0; k.nlaunch = k.st[s].end - k.wid0; k.stage_lo = s;
        hipLaunchKernelGGL(head_fwd_fused_kernel, dim3(k.nlaunch), dim3(256), 0, sm, k);
    }
    return pngpd_launch_status();
}

int pngpd_head_bwd_fused(const pngpd_head_train_t *a, const float *z1, const float *y1, const float *mean1,
                         const float *var1, const float *z2, const float *y2, const float *mean2, const float *var2,
                         float *dl, float *dy2, float *dz2, float *dy1, float *dz1, void *stream) {
    HeadBwdK k{};
    const bool lsm = a->epilogue == PNGPD_EPI_LOG_SOFTMAX;
    const int nrb = cdiv(a->B, 32), kb2 = cdiv(a->H2, 32), kb1 = cdiv(a->H1, 32), kb0 = cdiv(a->K0, 32);
    const int n0 = lsm ? cdiv(a->B, 256) : 0;
    const int e1 = n0 + nrb * kb2 + cdiv(a->k, 32) * kb2;
    const int e2 = e1 + nrb * kb1 + cdiv(a->H2, 32) * kb1;
    const int tx3 = a->dinp ? nrb * kb0 : 0;
    const int e3 = e2 + tx3 + cdiv(a->H1, 32) * kb0;
    k.st[0] = BwdStage{lsm ? dl : a->gout, y2, a->W3, a->dW3, dy2, a->db3, z2, y2, a->g2, mean2, var2, dz2, a->dg2,
                       a->dbe2, a->H2, a->k, 0, nrb * kb2, e1, TK_COLS, TK_DONE_A, TK_DONE0, n0, 1};
    k.st[1] = BwdStage{dz2, y1, a->W2, a->dW2, dy1, a->db2, z1, y1, a->g1, mean1, var1, dz1, a->dg1, a->dbe1,
                       a->H1, a->H2, 1, nrb * kb1, e2, TK_COLS + kb2, TK_DONE_B, TK_DONE_A, kb2, 1};
    k.st[2] = BwdStage{dz1, a->inp, a->W1, a->dW1, a->dinp, a->db1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr, nullptr, a->K0, a->H1, 1, tx3, e3, 0, 0, TK_DONE_B, kb1, 0};
    k.gout = a->gout; k.logp = a->out; k.dl = dl;
    k.B = a->B; k.k = a->k; k.nrb = nrb; k.n0 = n0; k.eps = a->eps; k.tk = a->tickets;
    k.nwords = (int)pngpd_head_train_ticket_ints(a);
    hipStream_t sm = (hipStream_t)stream;
    if (e3 <= resident_capacity((const void *)head_bwd_fused_kernel, 1)) {
        k.wid0 = 0; k.nlaunch = e3; k.stage_lo = 0;
        hipLaunchKernelGGL(head_bwd_fused_kernel, dim3(e3), dim3(256), 0, sm, k);
        return pngpd_launch_status();
    }
    const int ends[4] = {n0, e1, e2, e3};
    for (int s = 0; s < 4; ++s) {
        k.wid0 = s ? ends[s - 1] : 0; k.nlaunch = ends[s] - k.wid0; k.stage_lo = s;
        if (k.nlaunch == 0) continue;
        hipLaunchKernelGGL(head_bwd_fused_kernel, dim3(k.nlaunch), dim3(256), 0, sm, k);
    }
    return pngpd_launch_status();
}
