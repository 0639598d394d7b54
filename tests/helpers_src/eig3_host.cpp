// Host build of csrc/pngpd_gpg_eig3.h for the CPU tests (g++ -ffp-contract=off): the same source the device compiles.
#include "pngpd_gpg_eig3.h"
extern "C" {
void eig3_batch(const double *M, long n, double *w, double *v, int *info) {
    for (long i = 0; i < n; ++i) info[i] = pn_dgeev_sym3(M + 9 * i, w + 3 * i, v + 9 * i);
}
// intermediate results, column-major like LAPACK's: H, Q after DGEHD2 + DORGHR; T, Z after DLAHQR
void eig3_stages(const double *M, double *H, double *Q, double *T, double *Z, double *wr) {
    double h[9], z[9], wi[3] = {0, 0, 0};
    int info = 0, scale[3] = {1, 2, 3};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) h[i + 3 * j] = M[3 * i + j];
    const int ihi = e3_dgebal_perm(h, scale);
    e3_hessenberg(ihi, h, z);
    for (int q = 0; q < 9; ++q) { H[q] = h[q]; Q[q] = z[q]; }
    for (int i = ihi + 1; i <= 3; ++i) wr[i - 1] = E3_H(i, i);
    e3_dlahqr(ihi, h, wr, wi, z, info);
    for (int q = 0; q < 9; ++q) { T[q] = h[q]; Z[q] = z[q]; }
}
void eig3_dlanv2(double *abcd, double *out) {
    int info = 0;
    e3_dlanv2(abcd[0], abcd[1], abcd[2], abcd[3], out[0], out[1], out[2], out[3], out[4], out[5], info);
}
void eig3_frames(const double *M, const double *nat, const double *pts, long n, double *frames, int *flags) {
    for (long i = 0; i < n; ++i) flags[i] = pn_gpg_local_frame(M + 9 * i, nat + 3 * i, pts + 3 * i, frames + 12 * i);
}
double eig3_nrm2(double a, double b, double c) { return e3_nrm2(a, b, c); }
}
