// libpngpd — the 3x3 eigen-decomposition of the GPG sampler's local frame on the device.
//
// Reference call site replaced: dex-net/src/dexnet/grasping/grasp_sampler.py:1493  `eigval, eigvec = np.linalg.eig(M)`
// with M = sum of n n^T over the r-ball (symmetric, positive semi-definite), followed by :1494-1506 (minor / normal /
// major axes, flipped against the cloud normal).  Which of +v / -v LAPACK returns for an eigenvector decides the
// direction of the minor axis, hence the order in which the rotation x offset sweep (:1524-1541) enumerates poses and
// which candidates exist: an analytic or Jacobi solver would produce a different (mirrored) candidate list.  So this is
// the algorithm numpy runs, restated for n = 3: LAPACK's DGEEV('N','V') — third-party to the reference (numpy 2.2 /
// OpenBLAS 0.3.29 = LAPACK 3.11 in this image), published algorithm, routine by routine:
//   DGEBAL('B')   isolate eigenvalues by row permutations (a symmetric block is already balanced: c == r, no scaling)
//   DGEHD2        one Householder reflector (DLARFG + DLARF right/left) to Hessenberg form
//   DORGHR        Q = diag(1, I - tau v v^T)
//   DLAHQR        Francis double-shift QR with the Ahues-Tisseur deflation test, Schur vectors accumulated in Z;
//                 DLANV2 standardises a 2x2 block that splits off
//   DTREVC3       back-substitution per eigenvalue (DLALN2's perturbed pivot), Z * X, scaling by 1 / max|.|
//   DGEBAK, then  1 / DNRM2 per column
// Every operation is an IEEE fp64 operation in LAPACK's order WITHOUT fused multiply-add (the including file must be
// compiled with contraction off: `#pragma clang fp contract(off)` / g++ -ffp-contract=off), so device and host builds
// of this header agree bit for bit; against the library itself the vectors agree to a few ulp (OpenBLAS's own
// kernels fuse and use x87 for DNRM2) with the same signs and the same eigenvalue order — checked on the CPU over
// millions of matrices (tests/test_gpg_eig3.py), stage by stage against the library's own routines.
// Matrices whose decomposition is decided by rounding (two eigenvalues equal to the last bits: exactly planar
// synthetic patches) have no stable answer in LAPACK either; `eig="lapack"` in gpg.py keeps the library call.
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define PN_E3 __host__ __device__ __forceinline__
#else
#define PN_E3 static inline
#endif

#define E3_EPS_P 2.220446049250313e-16        /* DLAMCH('P') = eps * base */
#define E3_EPS_E 1.1102230246251565e-16       /* DLAMCH('E') */
#define E3_SAFMIN 2.2250738585072014e-308     /* DLAMCH('S') */
#define E3_OVERFLOW 1.7976931348623157e308    /* DLAMCH('O') */

// info bits of pn_dgeev_sym3
#define E3_INFO_NOCONV 1      /* DLAHQR did not converge (LAPACK would go on to DLAQR0) */
#define E3_INFO_COMPLEX 2     /* a complex pair: numpy would return complex arrays */
#define E3_INFO_RANGE 4       /* max|M| outside [sqrt(safmin)/eps, its inverse]: DGEEV would rescale */

PN_E3 double e3_sign(double a, double b) {      // Fortran SIGN(a, b): |a| with the sign bit of b (-0.0 counts as negative)
    return copysign(fabs(a), b);
}

PN_E3 double e3_dlapy2(double x, double y) {
    const double xa = fabs(x), ya = fabs(y);
    const double w = xa > ya ? xa : ya, z = xa > ya ? ya : xa;
    if (z == 0.0 || w > E3_OVERFLOW) return w;
    const double q = z / w;
    return w * sqrt(1.0 + q * q);
}

// DNRM2 of two / three elements as the library evaluates it (x87 kernel: sum of squares and square root in extended
// precision, rounded once to fp64; checked equal to the correctly rounded value on 2e5 random vectors): the squares and
// their sum as an unevaluated hi + lo pair, the root with one Newton correction from the exact residual.
PN_E3 double e3_nrm2(double x0, double x1, double x2) {
    const double p0 = x0 * x0, p1 = x1 * x1, p2 = x2 * x2;
    const double e0 = fma(x0, x0, -p0), e1 = fma(x1, x1, -p1), e2 = fma(x2, x2, -p2);
    const double s1 = p0 + p1, b1 = s1 - p0, t1 = (p0 - (s1 - b1)) + (p1 - b1);
    const double s2 = s1 + p2, b2 = s2 - s1, t2 = (s1 - (s2 - b2)) + (p2 - b2);
    const double lo = ((t1 + t2) + (e0 + e1)) + e2;
    const double hi = s2 + lo, lo2 = lo - (hi - s2);
    if (hi == 0.0) return 0.0;
    const double r = sqrt(hi);
    const double d = fma(-r, r, hi) + lo2;
    return r + d / (2.0 * r);
}

// DLARFG(n, alpha, x, 1, tau) for n = 2 or 3 (x has n - 1 elements).
PN_E3 void e3_dlarfg(int n, double &alpha, double *x, double &tau) {
    if (n <= 1) { tau = 0.0; return; }
    // DNRM2 of one or two elements (the library computes it in extended precision: exact for one element)
    double xnorm = n == 2 ? fabs(x[0]) : e3_nrm2(x[0], x[1], 0.0);
    if (xnorm == 0.0) { tau = 0.0; return; }
    double beta = -e3_sign(e3_dlapy2(alpha, xnorm), alpha);
    const double safmin = E3_SAFMIN / E3_EPS_E;
    int knt = 0;
    if (fabs(beta) < safmin) {
        const double rsafmn = 1.0 / safmin;
        do {
            ++knt;
            for (int i = 0; i < n - 1; ++i) x[i] = rsafmn * x[i];
            beta = beta * rsafmn;
            alpha = alpha * rsafmn;
        } while (fabs(beta) < safmin && knt < 20);
        xnorm = n == 2 ? fabs(x[0]) : e3_nrm2(x[0], x[1], 0.0);
        beta = -e3_sign(e3_dlapy2(alpha, xnorm), alpha);
    }
    tau = (beta - alpha) / beta;
    const double sc = 1.0 / (alpha - beta);
    for (int i = 0; i < n - 1; ++i) x[i] = sc * x[i];
    for (int j = 0; j < knt; ++j) beta = beta * safmin;
    alpha = beta;
}

// DLANV2: Schur factorisation of a real 2x2 block in standardised form.
PN_E3 void e3_dlanv2(double &a, double &b, double &c, double &d, double &rt1r, double &rt1i, double &rt2r, double &rt2i,
                     double &cs, double &sn, int &info) {
    const double multpl = 4.0, eps = E3_EPS_P;
    const double safmn2 = 1.0010415475915505e-146;   // 2^-485 = base^int(log(safmin/eps)/log(base)/2)
    const double safmx2 = 1.0 / safmn2;
    if (c == 0.0) {
        cs = 1.0; sn = 0.0;
    } else if (b == 0.0) {
        cs = 0.0; sn = 1.0;
        const double t = d; d = a; a = t; b = -c; c = 0.0;
    } else if ((a - d) == 0.0 && e3_sign(1.0, b) != e3_sign(1.0, c)) {
        cs = 1.0; sn = 0.0;
    } else {
        double temp = a - d;
        double p = 0.5 * temp;
        const double bcmax = fmax(fabs(b), fabs(c));
        const double bcmis = fmin(fabs(b), fabs(c)) * e3_sign(1.0, b) * e3_sign(1.0, c);
        double scale = fmax(fabs(p), bcmax);
        double z = (p / scale) * p + (bcmax / scale) * bcmis;
        if (z >= multpl * eps) {
            z = p + e3_sign(sqrt(scale) * sqrt(z), p);
            a = d + z;
            d = d - (bcmax / z) * bcmis;
            const double tau = e3_dlapy2(c, z);
            cs = z / tau;
            sn = c / tau;
            b = b - c;
            c = 0.0;
        } else {
            int count = 0;
            double sigma = b + c;
            bool done = false;
            cs = 1.0; sn = 0.0;
            while (!done) {
                ++count;
                scale = fmax(fabs(temp), fabs(sigma));
                if (scale >= safmx2) {
                    sigma = sigma * safmn2; temp = temp * safmn2;
                    if (count <= 20) continue;
                    info |= E3_INFO_NOCONV; break;
                } else if (scale <= safmn2) {
                    sigma = sigma * safmx2; temp = temp * safmx2;
                    if (count <= 20) continue;
                    info |= E3_INFO_NOCONV; break;
                }
                done = true;
                p = 0.5 * temp;
                double tau = e3_dlapy2(sigma, temp);
                cs = sqrt(0.5 * (1.0 + fabs(sigma) / tau));
                sn = -(p / (tau * cs)) * e3_sign(1.0, sigma);
                const double aa = a * cs + b * sn, bb = -a * sn + b * cs;
                const double cc = c * cs + d * sn, dd = -c * sn + d * cs;
                a = aa * cs + cc * sn;
                b = bb * cs + dd * sn;
                c = -aa * sn + cc * cs;
                d = -bb * sn + dd * cs;
                temp = 0.5 * (a + d);
                a = temp; d = temp;
                if (c != 0.0) {
                    if (b != 0.0) {
                        if (e3_sign(1.0, b) == e3_sign(1.0, c)) {
                            const double sab = sqrt(fabs(b)), sac = sqrt(fabs(c));
                            p = e3_sign(sab * sac, c);
                            tau = 1.0 / sqrt(fabs(b + c));
                            a = temp + p;
                            d = temp - p;
                            b = b - c;
                            c = 0.0;
                            const double cs1 = sab * tau, sn1 = sac * tau;
                            temp = cs * cs1 - sn * sn1;
                            sn = cs * sn1 + sn * cs1;
                            cs = temp;
                        }
                    } else {
                        b = -c; c = 0.0;
                        temp = cs; cs = -sn; sn = temp;
                    }
                }
            }
        }
    }
    rt1r = a; rt2r = d;
    if (c == 0.0) {
        rt1i = 0.0; rt2i = 0.0;
    } else {
        rt1i = sqrt(fabs(b)) * sqrt(fabs(c));
        rt2i = -rt1i;
    }
}

#define E3_H(i, j) h[((i) - 1) + 3 * ((j) - 1)]
#define E3_Z(i, j) z[((i) - 1) + 3 * ((j) - 1)]

// DGEBAL('B') of a 3x3 matrix, permutation part (h column-major, in place).  Returns ihi (ilo is 1 for a symmetric
// matrix: a column is isolated exactly when its row is, and the rows went first); scale[j - 1] for j > ihi = the row j
// was exchanged with (1-based).
PN_E3 int e3_dgebal_perm(double *h, int *scale) {
    const int n = 3, k = 1;
    int l = n;
    bool noconv = true;
    while (noconv) {
        noconv = false;
        for (int i = l; i >= 1; --i) {
            bool canswap = true;
            for (int j = 1; j <= l; ++j)
                if (i != j && E3_H(i, j) != 0.0) { canswap = false; break; }
            if (canswap) {
                scale[l - 1] = i;
                if (i != l) {
                    for (int r = 1; r <= l; ++r) { const double t = E3_H(r, i); E3_H(r, i) = E3_H(r, l); E3_H(r, l) = t; }
                    for (int cc = k; cc <= n; ++cc) { const double t = E3_H(i, cc); E3_H(i, cc) = E3_H(l, cc); E3_H(l, cc) = t; }
                }
                noconv = true;
                if (l == 1) return 1;
                l = l - 1;
            }
        }
    }
    // columns: nothing can be isolated in what is left of a symmetric matrix (see above); kept for a general input
    int kk = k;
    noconv = true;
    while (noconv) {
        noconv = false;
        for (int j = kk; j <= l; ++j) {
            bool canswap = true;
            for (int i = kk; i <= l; ++i)
                if (i != j && E3_H(i, j) != 0.0) { canswap = false; break; }
            if (canswap) {
                if (j != kk) return -1;       // would move ilo: not a symmetric input
                return -1;
            }
        }
    }
    return l;
}

// DLAHQR(wantt, wantz) on rows / columns 1..ihi of the 3x3 Hessenberg matrix h, Schur vectors accumulated into
// rows 1..ihi of z.  Returns 0 or the index at which it failed to converge.
PN_E3 int e3_dlahqr(int ihi, double *h, double *wr, double *wi, double *z, int &info) {
    const int n = 3, ilo = 1, iloz = 1, ihiz = ihi;
    const double dat1 = 0.75, dat2 = -0.4375;
    const int kexsh = 10;
    if (ilo == ihi) { wr[0] = E3_H(1, 1); wi[0] = 0.0; return 0; }
    if (ilo <= ihi - 2) E3_H(ihi, ihi - 2) = 0.0;
    const int nh = ihi - ilo + 1, nz = ihiz - iloz + 1;
    const double safmin = E3_SAFMIN, ulp = E3_EPS_P;
    const double smlnum = safmin * ((double)nh / ulp);
    const int i1 = 1, i2 = n;
    const int itmax = 30 * (nh > 10 ? nh : 10);
    int kdefl = 0;
    int i = ihi;
    while (true) {
        int l = ilo;
        if (i < ilo) return 0;
        bool converged = false;
        for (int its = 0; its <= itmax; ++its) {
            int k;
            for (k = i; k >= l + 1; --k) {
                if (fabs(E3_H(k, k - 1)) <= smlnum) break;
                double tst = fabs(E3_H(k - 1, k - 1)) + fabs(E3_H(k, k));
                if (tst == 0.0) {
                    if (k - 2 >= ilo) tst = tst + fabs(E3_H(k - 1, k - 2));
                    if (k + 1 <= ihi) tst = tst + fabs(E3_H(k + 1, k));
                }
                if (fabs(E3_H(k, k - 1)) <= ulp * tst) {
                    const double hkk1 = fabs(E3_H(k, k - 1)), hk1k = fabs(E3_H(k - 1, k));
                    const double ab = fmax(hkk1, hk1k), ba = fmin(hkk1, hk1k);
                    const double dkk = fabs(E3_H(k, k)), ddf = fabs(E3_H(k - 1, k - 1) - E3_H(k, k));
                    const double aa = fmax(dkk, ddf), bb = fmin(dkk, ddf);
                    const double s = aa + ab;
                    if (ba * (ab / s) <= fmax(smlnum, ulp * (bb * (aa / s)))) break;
                }
            }
            l = k;
            if (l > ilo) E3_H(l, l - 1) = 0.0;
            if (l >= i - 1) { converged = true; break; }
            kdefl = kdefl + 1;
            double h11, h21, h12, h22;
            if (kdefl % (2 * kexsh) == 0) {
                const double s = fabs(E3_H(i, i - 1)) + fabs(E3_H(i - 1, i - 2));
                h11 = dat1 * s + E3_H(i, i); h12 = dat2 * s; h21 = s; h22 = h11;
            } else if (kdefl % kexsh == 0) {
                const double s = fabs(E3_H(l + 1, l)) + fabs(E3_H(l + 2, l + 1));
                h11 = dat1 * s + E3_H(l, l); h12 = dat2 * s; h21 = s; h22 = h11;
            } else {
                h11 = E3_H(i - 1, i - 1); h21 = E3_H(i, i - 1); h12 = E3_H(i - 1, i); h22 = E3_H(i, i);
            }
            double s = fabs(h11) + fabs(h12) + fabs(h21) + fabs(h22);
            double rt1r, rt1i, rt2r, rt2i;
            if (s == 0.0) {
                rt1r = 0.0; rt1i = 0.0; rt2r = 0.0; rt2i = 0.0;
            } else {
                h11 = h11 / s; h21 = h21 / s; h12 = h12 / s; h22 = h22 / s;
                const double tr = (h11 + h22) / 2.0;
                const double det = (h11 - tr) * (h22 - tr) - h12 * h21;
                const double rtdisc = sqrt(fabs(det));
                if (det >= 0.0) {
                    rt1r = tr * s; rt2r = rt1r; rt1i = rtdisc * s; rt2i = -rt1i;
                } else {
                    rt1r = tr + rtdisc; rt2r = tr - rtdisc;
                    if (fabs(rt1r - h22) <= fabs(rt2r - h22)) { rt1r = rt1r * s; rt2r = rt1r; }
                    else { rt2r = rt2r * s; rt1r = rt2r; }
                    rt1i = 0.0; rt2i = 0.0;
                }
            }
            // two consecutive small subdiagonal elements: the active block is the whole 3x3 here (l = 1, i = 3), m = l
            int m;
            double v[3];
            for (m = i - 2; m >= l; --m) {
                double h21s = fabs(E3_H(m + 1, m));
                s = fabs(E3_H(m, m) - rt2r) + fabs(rt2i) + h21s;
                h21s = E3_H(m + 1, m) / s;
                v[0] = h21s * E3_H(m, m + 1) + (E3_H(m, m) - rt1r) * ((E3_H(m, m) - rt2r) / s) - rt1i * (rt2i / s);
                v[1] = h21s * (E3_H(m, m) + E3_H(m + 1, m + 1) - rt1r - rt2r);
                v[2] = h21s * E3_H(m + 2, m + 1);
                s = fabs(v[0]) + fabs(v[1]) + fabs(v[2]);
                v[0] = v[0] / s; v[1] = v[1] / s; v[2] = v[2] / s;
                if (m == l) break;
                const double h00 = fabs(E3_H(m - 1, m - 1)), h11a = fabs(E3_H(m, m)), h22a = fabs(E3_H(m + 1, m + 1));
                if (fabs(E3_H(m, m - 1)) * (fabs(v[1]) + fabs(v[2])) <= ulp * fabs(v[0]) * (h00 + h11a + h22a)) break;
            }
            for (k = m; k <= i - 1; ++k) {
                const int nr = (3 < i - k + 1) ? 3 : (i - k + 1);
                if (k > m) for (int q = 0; q < nr; ++q) v[q] = E3_H(k + q, k - 1);
                double t1;
                e3_dlarfg(nr, v[0], &v[1], t1);
                if (k > m) {
                    E3_H(k, k - 1) = v[0];
                    E3_H(k + 1, k - 1) = 0.0;
                    if (k < i - 1) E3_H(k + 2, k - 1) = 0.0;
                } else if (m > l) {
                    E3_H(k, k - 1) = E3_H(k, k - 1) * (1.0 - t1);
                }
                const double v2 = v[1], t2 = t1 * v2;
                if (nr == 3) {
                    const double v3 = v[2], t3 = t1 * v3;
                    for (int j = k; j <= i2; ++j) {
                        const double sum = E3_H(k, j) + v2 * E3_H(k + 1, j) + v3 * E3_H(k + 2, j);
                        E3_H(k, j) = E3_H(k, j) - sum * t1;
                        E3_H(k + 1, j) = E3_H(k + 1, j) - sum * t2;
                        E3_H(k + 2, j) = E3_H(k + 2, j) - sum * t3;
                    }
                    const int jhi = (k + 3 < i) ? k + 3 : i;
                    for (int j = i1; j <= jhi; ++j) {
                        const double sum = E3_H(j, k) + v2 * E3_H(j, k + 1) + v3 * E3_H(j, k + 2);
                        E3_H(j, k) = E3_H(j, k) - sum * t1;
                        E3_H(j, k + 1) = E3_H(j, k + 1) - sum * t2;
                        E3_H(j, k + 2) = E3_H(j, k + 2) - sum * t3;
                    }
                    for (int j = iloz; j <= ihiz; ++j) {
                        const double sum = E3_Z(j, k) + v2 * E3_Z(j, k + 1) + v3 * E3_Z(j, k + 2);
                        E3_Z(j, k) = E3_Z(j, k) - sum * t1;
                        E3_Z(j, k + 1) = E3_Z(j, k + 1) - sum * t2;
                        E3_Z(j, k + 2) = E3_Z(j, k + 2) - sum * t3;
                    }
                } else if (nr == 2) {
                    for (int j = k; j <= i2; ++j) {
                        const double sum = E3_H(k, j) + v2 * E3_H(k + 1, j);
                        E3_H(k, j) = E3_H(k, j) - sum * t1;
                        E3_H(k + 1, j) = E3_H(k + 1, j) - sum * t2;
                    }
                    for (int j = i1; j <= i; ++j) {
                        const double sum = E3_H(j, k) + v2 * E3_H(j, k + 1);
                        E3_H(j, k) = E3_H(j, k) - sum * t1;
                        E3_H(j, k + 1) = E3_H(j, k + 1) - sum * t2;
                    }
                    for (int j = iloz; j <= ihiz; ++j) {
                        const double sum = E3_Z(j, k) + v2 * E3_Z(j, k + 1);
                        E3_Z(j, k) = E3_Z(j, k) - sum * t1;
                        E3_Z(j, k + 1) = E3_Z(j, k + 1) - sum * t2;
                    }
                }
            }
        }
        if (!converged) return i;
        if (l == i) {
            wr[i - 1] = E3_H(i, i); wi[i - 1] = 0.0;
        } else if (l == i - 1) {
            double cs, sn;
            e3_dlanv2(E3_H(i - 1, i - 1), E3_H(i - 1, i), E3_H(i, i - 1), E3_H(i, i), wr[i - 2], wi[i - 2], wr[i - 1], wi[i - 1],
                      cs, sn, info);
            // DROT over the rest of H (rows i-1, i right of the block; columns i-1, i above it) and over Z, rounded as
            // the library's kernel rounds it: x' = fma(c, x, s y), y' = fma(c, y, -(s x))
            for (int j = i + 1; j <= i2; ++j) {
                const double x = E3_H(i - 1, j), y = E3_H(i, j);
                E3_H(i - 1, j) = fma(cs, x, sn * y);
                E3_H(i, j) = fma(cs, y, -(sn * x));
            }
            for (int j = i1; j <= i - 2; ++j) {
                const double x = E3_H(j, i - 1), y = E3_H(j, i);
                E3_H(j, i - 1) = fma(cs, x, sn * y);
                E3_H(j, i) = fma(cs, y, -(sn * x));
            }
            for (int j = iloz; j < iloz + nz; ++j) {
                const double x = E3_Z(j, i - 1), y = E3_Z(j, i);
                E3_Z(j, i - 1) = fma(cs, x, sn * y);
                E3_Z(j, i) = fma(cs, y, -(sn * x));
            }
        }
        kdefl = 0;
        i = l - 1;
    }
}

// DTREVC3('R', 'B') for a real upper-triangular t (all eigenvalues real), blocked form: X (upper triangular,
// unit diagonal) by back-substitution, then VR := Z * X, every column scaled by 1 / max|.|.
PN_E3 void e3_dtrevc3(const double *t, double *z) {
    const int n = 3;
    const double unfl = E3_SAFMIN, ulp = E3_EPS_P;
    const double smlnum = unfl * ((double)n / ulp), bignum = (1.0 - ulp) / smlnum;
#define E3_T(i, j) t[((i) - 1) + 3 * ((j) - 1)]
    double cn[3];                                   // 1-norms of the strictly upper columns
    cn[0] = 0.0;
    cn[1] = fabs(E3_T(1, 2));
    cn[2] = fabs(E3_T(1, 3)) + fabs(E3_T(2, 3));
    double x[9];                                    // x[(j-1) + 3 (ki-1)]
    for (int ki = n; ki >= 1; --ki) {
        const double wr = E3_T(ki, ki);
        const double smin = fmax(ulp * fabs(wr), smlnum);
        double *w = &x[3 * (ki - 1)];
        w[ki - 1] = 1.0;
        for (int k = 1; k <= ki - 1; ++k) w[k - 1] = -E3_T(k, ki);
        for (int k = ki + 1; k <= n; ++k) w[k - 1] = 0.0;
        for (int j = ki - 1; j >= 1; --j) {
            // DLALN2(na = 1, nw = 1): (t(j,j) - wr) x = scale * w(j), pivot perturbed to smini when smaller
            const double smini = fmax(smin, 2.0 * E3_SAFMIN);
            double csr = 1.0 * E3_T(j, j) - wr * 1.0;
            double cnorm = fabs(csr);
            if (cnorm < smini) { csr = smini; cnorm = smini; }
            const double bnorm = fabs(w[j - 1]);
            double scale = 1.0;
            const double big = 1.0 / (2.0 * E3_SAFMIN);
            if (cnorm < 1.0 && bnorm > 1.0) {
                if (bnorm > big * cnorm) scale = 1.0 / bnorm;
            }
            double xj = (w[j - 1] * scale) / csr;
            const double xnorm = fabs(xj);
            if (xnorm > 1.0) {
                if (cn[j - 1] > bignum / xnorm) { xj = xj / xnorm; scale = scale / xnorm; }
            }
            if (scale != 1.0) for (int q = 0; q < ki; ++q) w[q] = scale * w[q];
            w[j - 1] = xj;
            for (int q = 1; q <= j - 1; ++q) w[q - 1] = w[q - 1] + (-xj) * E3_T(q, j);      // DAXPY
        }
    }
    // DGEMM: VR = Z * X (k ascending), then the scaling
    double vr[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            double acc = z[r + 0] * x[0 + 3 * c];
            acc = acc + z[r + 3] * x[1 + 3 * c];
            acc = acc + z[r + 6] * x[2 + 3 * c];
            vr[r + 3 * c] = acc;
        }
    for (int c = 0; c < 3; ++c) {
        int ii = 0;                                  // IDAMAX: first maximum
        double mx = fabs(vr[3 * c]);
        for (int r = 1; r < 3; ++r) if (fabs(vr[r + 3 * c]) > mx) { mx = fabs(vr[r + 3 * c]); ii = r; }
        const double remax = 1.0 / fabs(vr[ii + 3 * c]);
        for (int r = 0; r < 3; ++r) z[r + 3 * c] = remax * vr[r + 3 * c];
    }
#undef E3_T
}

// DGEHD2 + DORGHR (ilo = 1): h -> upper Hessenberg in place, z = Q.
PN_E3 void e3_hessenberg(int ihi, double *h, double *z) {
    // DGEHD2 (ilo = 1): one reflector when ihi = 3
    double tau = 0.0, v2 = 0.0;
    if (ihi == 3) {
        double alpha = E3_H(2, 1), x = E3_H(3, 1);
        e3_dlarfg(2, alpha, &x, tau);
        v2 = x;
        E3_H(2, 1) = alpha;          // beta; the vector element lives in v2 (LAPACK keeps it in H(3,1))
        E3_H(3, 1) = 0.0;
        if (tau != 0.0) {
            // DLARF('Right', 3, 2): C = H(1:3, 2:3);  w = C v;  C -= tau w v^T
            double wv[3];
            // (the library's DGEMV-N and DGER kernels fuse multiply and add, its DGEMV-T does not: determined bit for
            // bit against OpenBLAS 0.3.29, tests/test_gpg_eig3.py)
            for (int r = 1; r <= 3; ++r) wv[r - 1] = fma(v2, E3_H(r, 3), E3_H(r, 2));
            {
                const double t1 = -tau * 1.0, t2 = -tau * v2;
                for (int r = 1; r <= 3; ++r) E3_H(r, 2) = fma(wv[r - 1], t1, E3_H(r, 2));
                for (int r = 1; r <= 3; ++r) E3_H(r, 3) = fma(wv[r - 1], t2, E3_H(r, 3));
            }
            // DLARF('Left', 2, 2): C = H(2:3, 2:3);  w = C^T v;  C -= tau v w^T
            double wl[2];
            for (int c = 2; c <= 3; ++c) {
                double temp = 0.0;
                temp = temp + E3_H(2, c) * 1.0;
                temp = temp + E3_H(3, c) * v2;
                wl[c - 2] = 1.0 * temp;
            }
            for (int c = 2; c <= 3; ++c) {
                const double temp = -tau * wl[c - 2];
                E3_H(2, c) = E3_H(2, c) + temp;
                E3_H(3, c) = fma(v2, temp, E3_H(3, c));
            }
        }
    }
    // DORGHR: Q = diag(1, I - tau v v^T) exactly as DORG2R builds it
    for (int q = 0; q < 9; ++q) z[q] = 0.0;
    E3_Z(1, 1) = 1.0; E3_Z(2, 2) = 1.0; E3_Z(3, 3) = 1.0;
    if (ihi == 3) {
        if (tau != 0.0) {
            const double temp = -tau * v2;          // DGER's alpha * y(1) with w = v2
            E3_Z(2, 3) = 0.0 + 1.0 * temp;
            E3_Z(3, 3) = fma(v2, temp, 1.0);
        }
        E3_Z(3, 2) = -tau * v2;                     // DSCAL(1, -tau, A(2,1))
        E3_Z(2, 2) = 1.0 - tau;
    }
}

// np.linalg.eig of a symmetric 3x3 matrix, M row-major as numpy holds it -> w[3], v[9] ROW-major like numpy's result
// (column k of v = eigenvector of w[k]).  Returns info (0 = as LAPACK's real-eigenvalue path).
PN_E3 int pn_dgeev_sym3(const double *M, double *w, double *v) {
    int info = 0;
    double h[9], z[9], wi[3] = {0.0, 0.0, 0.0};
    double anrm = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            h[i + 3 * j] = M[3 * i + j];
            const double a = fabs(M[3 * i + j]);
            if (a > anrm || a != a) anrm = a;
        }
    const double smlnum = sqrt(E3_SAFMIN) / E3_EPS_P, bignum = 1.0 / smlnum;
    if ((anrm > 0.0 && anrm < smlnum) || anrm > bignum || anrm != anrm) info |= E3_INFO_RANGE;
    int scale[3] = {1, 2, 3};
    const int ihi = e3_dgebal_perm(h, scale);
    if (ihi < 0) { info |= E3_INFO_RANGE; w[0] = w[1] = w[2] = 0.0; for (int q = 0; q < 9; ++q) v[q] = 0.0; return info; }
    e3_hessenberg(ihi, h, z);
    // DHSEQR -> DLAHQR on 1..ihi; isolated eigenvalues are the diagonal entries
    for (int i = ihi + 1; i <= 3; ++i) { w[i - 1] = E3_H(i, i); wi[i - 1] = 0.0; }
    const int fail = e3_dlahqr(ihi, h, w, wi, z, info);
    if (fail) info |= E3_INFO_NOCONV;
    if (wi[0] != 0.0 || wi[1] != 0.0 || wi[2] != 0.0) info |= E3_INFO_COMPLEX;
    if (ihi == 3) E3_H(3, 1) = 0.0;
    e3_dtrevc3(h, z);
    // DGEBAK('B', 'R'): rows ihi+1..3 back in place
    for (int i = ihi + 1; i <= 3; ++i) {
        const int k = scale[i - 1];
        if (k == i) continue;
        for (int c = 1; c <= 3; ++c) { const double t = E3_Z(i, c); E3_Z(i, c) = E3_Z(k, c); E3_Z(k, c) = t; }
    }
    // 1 / DNRM2 per column
    for (int c = 1; c <= 3; ++c) {
        const double nrm = e3_nrm2(E3_Z(1, c), E3_Z(2, c), E3_Z(3, c));
        const double scl = 1.0 / nrm;
        for (int r = 1; r <= 3; ++r) v[3 * (r - 1) + (c - 1)] = scl * E3_Z(r, c);
    }
    return info;
}

// grasp_sampler.py:1486-1506 for one sample point: m = its moment matrix (row-major), na = all_normal[ind], pt = the
// sample point -> f[12] = minor_pc, new_normal, major_pc, pt; returns the frame flags of pngpd_gpg_frames (pngpd.h).
// numpy's roundings: separate multiply / add, sums left to right, x / |x| by division.  A point whose M sums to zero
// (:1486 `continue`) gets the frame of a hand parked GPG_FAR metres away: no cloud point lies in any of its boxes, so it
// yields no grasp and a round needs no compaction (and no host round trip) between the moments and the sweep.
#define GPG_FAR 1.0e6
#define GPG_FRAME_DEAD 1
#define GPG_FRAME_COMPLEX 2
#define GPG_FRAME_NOCONV 4
#define GPG_FRAME_RANGE 8

PN_E3 void e3_unit_col(const double *v, int col, double *o) {      // v row-major (3,3): column `col`, normalised
    const double a = v[col], b = v[3 + col], c = v[6 + col];
    const double n = sqrt((a * a + b * b) + c * c);
    o[0] = a / n; o[1] = b / n; o[2] = c / n;
}

PN_E3 int pn_gpg_local_frame(const double *m, const double *na, const double *pt, double *f) {
    // sum(sum(M)): python's sum over the rows (column sums, top to bottom), then over those
    const double c0 = (m[0] + m[3]) + m[6], c1 = (m[1] + m[4]) + m[7], c2 = (m[2] + m[5]) + m[8];
    if ((c0 + c1) + c2 == 0.0) {
        f[0] = 1.0; f[1] = 0.0; f[2] = 0.0; f[3] = 0.0; f[4] = 1.0; f[5] = 0.0; f[6] = 0.0; f[7] = 0.0; f[8] = 1.0;
        f[9] = GPG_FAR; f[10] = GPG_FAR; f[11] = GPG_FAR;
        return GPG_FRAME_DEAD;
    }
    double w[3], v[9];
    const int info = pn_dgeev_sym3(m, w, v);
    int imin = 0, imax = 0;                                        // np.argmin / np.argmax: the first extreme value
    if (w[1] < w[imin]) imin = 1;
    if (w[2] < w[imin]) imin = 2;
    if (w[1] > w[imax]) imax = 1;
    if (w[2] > w[imax]) imax = 2;
    double minor[3], normal[3], major[3];
    e3_unit_col(v, imin, minor);
    e3_unit_col(v, imax, normal);
    major[0] = minor[1] * normal[2] - minor[2] * normal[1];
    major[1] = minor[2] * normal[0] - minor[0] * normal[2];
    major[2] = minor[0] * normal[1] - minor[1] * normal[0];
    const double nm = sqrt((major[0] * major[0] + major[1] * major[1]) + major[2] * major[2]);
    if (nm != 0.0) { major[0] = major[0] / nm; major[1] = major[1] / nm; major[2] = major[2] / nm; }
    const double dt = (na[0] * normal[0] + na[1] * normal[1]) + na[2] * normal[2];
    if (dt < 0.0)
        for (int i = 0; i < 3; ++i) { normal[i] = -normal[i]; minor[i] = -minor[i]; }
    for (int i = 0; i < 3; ++i) { f[i] = minor[i]; f[3 + i] = normal[i]; f[6 + i] = major[i]; f[9 + i] = pt[i]; }
    return ((info & E3_INFO_COMPLEX) ? GPG_FRAME_COMPLEX : 0) | ((info & E3_INFO_NOCONV) ? GPG_FRAME_NOCONV : 0) |
           ((info & E3_INFO_RANGE) ? GPG_FRAME_RANGE : 0);
}
