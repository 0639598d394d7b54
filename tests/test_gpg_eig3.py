"""csrc/pngpd_gpg_eig3.h — LAPACK's DGEEV restated for the sampler's symmetric 3x3 moment matrices — against numpy's own
LAPACK (the library `np.linalg.eig` calls at grasp_sampler.py:1493).  The header is compiled for the HOST here (g++,
contraction off: the same source and the same roundings the device build has; tests/test_gpu_gpg.py checks the kernel
against this build bit for bit), so every statement below is about the arithmetic the GPU runs.

What must hold: same eigenvalue ORDER and same eigenvector SIGNS as the library (they decide the sweep's enumeration order
and which candidates exist), values to a few ulp, and the library's complex-pair cases flagged exactly.  Matrices whose QR
iteration sits within rounding of a deflation threshold are sign-unstable in LAPACK ITSELF (a 1e-13 relative perturbation of
the input flips them): the restatement follows the library bit for bit through the reduction and the QR sweeps, so that even
those agree on the CPU family the fixtures were generated on (counted below, not assumed)."""
import ctypes
import glob
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "pointnetgpd_amd", "csrc")


@pytest.fixture(scope="module")
def e3(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("eig3") / "libeig3_host.so")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I", CSRC,
                    os.path.join(HERE, "helpers_src", "eig3_host.cpp"), "-o", out], check=True)
    return ctypes.CDLL(out)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def eig3(lib, M):
    M = np.ascontiguousarray(M, dtype=np.float64)
    n = M.shape[0]
    w, v, info = np.empty((n, 3)), np.empty((n, 3, 3)), np.empty(n, dtype=np.int32)
    lib.eig3_batch(_p(M), ctypes.c_long(n), _p(w), _p(v), _p(info))
    return w, v, info


def moment_matrices(rng, n, kmin=2, kmax=100, noise=0.05):
    """M = sum n n^T exactly as grasp_sampler.py:1477-1484 accumulates it: 1..100 unit normals scattered around a patch
    normal (noise 0.01 = a flat face seen by a good sensor ... 0.5 = an edge / clutter)."""
    out = np.zeros((n, 3, 3))
    base = rng.normal(size=(n, 3))
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    cnt = rng.integers(kmin, kmax + 1, n)
    for i in range(n):
        nr = base[i] + rng.normal(scale=noise, size=(cnt[i], 3))
        nr /= np.linalg.norm(nr, axis=1, keepdims=True)
        M = np.zeros((3, 3))
        for q in nr:
            M += np.matmul(q.reshape(3, 1), q.reshape(1, 3))
        out[i] = M
    return out


def _openblas_arch():
    from threadpoolctl import threadpool_info
    arch = [d.get("architecture") for d in threadpool_info() if d.get("internal_api") == "openblas"]
    return arch[0] if arch else None


def lapack_each(M):
    W = np.zeros((len(M), 3), dtype=complex)
    V = np.zeros((len(M), 3, 3), dtype=complex)
    for i in range(len(M)):
        W[i], V[i] = np.linalg.eig(M[i])
    return W, V


@pytest.mark.parametrize("noise,kmin", [(0.05, 1), (0.05, 2), (0.2, 2), (0.01, 2), (0.5, 2)])
def test_matches_numpy_eig_order_signs_values(e3, noise, kmin):
    rng = np.random.default_rng(int(noise * 1000) + kmin)
    M = moment_matrices(rng, 6000, kmin=kmin, noise=noise)
    W, V = lapack_each(M)
    w, v, info = eig3(e3, M)
    cplx = np.abs(W.imag).max(1) > 0
    # the library's complex pairs (rank-1 M: one neighbour) are flagged, exactly those
    np.testing.assert_array_equal((info & 2) != 0, cplx)
    assert not (info & ~2).any()
    real = ~cplx
    dv = np.abs(V.real - v).max((1, 2))
    agree = dv[real] < 1e-9
    # sign / order agreement: every matrix but (at most) the rounding-decided ones.  On another OpenBLAS kernel family the
    # LIBRARY's own roundings differ in the last bit and with them its verdict on the rounding-decided ~3 % (measured before
    # the header followed the SkylakeX kernels' fused / unfused pattern: 96.5 %)
    need = 0.9995 if _openblas_arch() == "SkylakeX" else 0.95
    assert agree.mean() >= need, f"{(~agree).sum()} of {real.sum()} differ in sign or order"
    assert np.abs(W.real - w)[real].max() < 1e-12 * max(1.0, np.abs(w).max())
    assert dv[real][agree].max() < 1e-11
    bit = (np.all(V.real == v, axis=(1, 2)) & np.all(W.real == w, axis=1))[real].mean()
    print(f"noise {noise} kmin {kmin}: complex {cplx.sum()}, sign/order agreement {agree.mean():.5f}, bit-identical {bit:.4f}")


def test_sparsity_patterns_and_isolated_eigenvalues(e3):
    """DGEBAL's permutations: every zero pattern of the off-diagonal, diagonal matrices, repeated entries."""
    rng = np.random.default_rng(7)
    mats = []
    for mask in range(8):
        for _ in range(40):
            a = rng.uniform(0.1, 50, 3)
            o = rng.normal(size=3) * [(mask >> k) & 1 for k in range(3)]
            mats.append(np.array([[a[0], o[0], o[1]], [o[0], a[1], o[2]], [o[1], o[2], a[2]]]))
    mats += [np.diag([3.0, 1.0, 2.0]), np.diag([1.0, 1.0, 2.0]), np.diag([0.0, 0.0, 5.0]), np.eye(3) * 4.0,
             np.array([[2.0, 1, 0], [1, 2, 0], [0, 0, 2.0]]), np.array([[2.0, 0, 1], [0, 2, 0], [1, 0, 2.0]])]
    M = np.array(mats)
    W, V = lapack_each(M)
    w, v, info = eig3(e3, M)
    assert not np.abs(W.imag).any() and not info.any()
    np.testing.assert_allclose(w, W.real, rtol=0, atol=1e-13)
    np.testing.assert_allclose(v, V.real, rtol=0, atol=1e-13)


def _numpy_openblas():
    import numpy
    hits = glob.glob(os.path.join(os.path.dirname(numpy.__file__), "..", "numpy.libs", "libscipy_openblas64_*.so"))
    if not hits:
        return None
    L = ctypes.CDLL(hits[0])
    return L if hasattr(L, "scipy_dgehrd_64_") and hasattr(L, "scipy_dlahqr_64_") else None


def test_stages_against_the_library_routines(e3):
    """The reduction (DGEBAL + DGEHD2 + DORGHR) and the QR iteration (DLAHQR), each against the routine numpy's own
    OpenBLAS exports, on identical inputs: Hessenberg form and Q bit for bit, Schur form and vectors with the same signs."""
    L = _numpy_openblas()
    if L is None:
        pytest.skip("numpy's bundled OpenBLAS (ILP64, scipy_ prefix) not found")
    i64 = ctypes.c_int64

    def I(v):
        return ctypes.byref(i64(v))

    rng = np.random.default_rng(11)
    M = moment_matrices(rng, 1500, noise=0.1)
    F = lambda x: x.reshape(3, 3).T.copy()
    h_bit = z_sign = z_bit = 0
    for A in M:
        H, Q, T, Z, W = np.zeros(9), np.zeros(9), np.zeros(9), np.zeros(9), np.zeros(3)
        e3.eig3_stages(_p(np.ascontiguousarray(A)), _p(H), _p(Q), _p(T), _p(Z), _p(W))
        a = np.asfortranarray(A.copy())
        tau, work, info = np.zeros(2), np.zeros(256), i64()
        L.scipy_dgehrd_64_(I(3), I(1), I(3), _p(a), I(3), _p(tau), _p(work), I(256), ctypes.byref(info))
        q = np.asfortranarray(np.tril(a))
        L.scipy_dorghr_64_(I(3), I(1), I(3), _p(q), I(3), _p(tau), _p(work), I(256), ctypes.byref(info))
        h_bit += np.array_equal(np.triu(a, -1), F(H)) and np.array_equal(q, F(Q))
        h2, z2 = np.asfortranarray(F(H)), np.asfortranarray(F(Q))
        wr, wi, one = np.zeros(3), np.zeros(3), i64(1)
        L.scipy_dlahqr_64_(ctypes.byref(one), ctypes.byref(one), I(3), I(1), I(3), _p(h2), I(3), _p(wr), _p(wi), I(1), I(3),
                           _p(z2), I(3), ctypes.byref(info))
        z_sign += np.abs(z2 - F(Z)).max() < 1e-9
        z_bit += np.array_equal(z2, F(Z))
    n = len(M)
    print(f"Hessenberg + Q bit-identical {h_bit}/{n}; DLAHQR same signs {z_sign}/{n}, bit-identical {z_bit}/{n}")
    assert z_sign >= n - 2
    if _openblas_arch() == "SkylakeX":      # the kernel family whose fused / unfused roundings the header follows
        assert h_bit == n
        assert z_bit >= n - 5


def test_dlanv2_and_dnrm2_against_the_library(e3):
    L = _numpy_openblas()
    if L is None:
        pytest.skip("numpy's bundled OpenBLAS (ILP64, scipy_ prefix) not found")
    dbl = ctypes.c_double
    L.scipy_dnrm2_64_.restype = dbl
    e3.eig3_nrm2.restype = dbl
    rng = np.random.default_rng(3)
    for it in range(20000):
        a, b, c, d = rng.normal(size=4)
        kind = it % 5
        if kind == 1:
            c = b * (1 + 1e-15 * rng.normal())
        elif kind == 2:
            c, d = b, a + 1e-14 * rng.normal()
        elif kind == 3:
            b, c = 1e-16 * rng.normal(), 1e-16 * rng.normal()
        elif kind == 4:
            b = c = 1e-9 * rng.normal()
            d = a * (1 + 1e-13)
        v = [dbl(x) for x in (a, b, c, d)]
        o = [dbl() for _ in range(6)]
        L.scipy_dlanv2_64_(*[ctypes.byref(x) for x in v], *[ctypes.byref(x) for x in o])
        ref = np.array([x.value for x in v] + [x.value for x in o])
        mine_in, mine_out = np.array([a, b, c, d]), np.zeros(6)
        e3.eig3_dlanv2(_p(mine_in), _p(mine_out))
        np.testing.assert_array_equal(np.concatenate([mine_in, mine_out]), ref)
    bad = 0
    for it in range(20000):
        x = rng.normal(size=3) * 10.0 ** rng.integers(-8, 3, size=3)
        n = 2 + it % 2
        if n == 2:
            x[2] = 0.0
        r = L.scipy_dnrm2_64_(ctypes.byref(ctypes.c_int64(n)), _p(x), ctypes.byref(ctypes.c_int64(1)))
        bad += r != e3.eig3_nrm2(dbl(x[0]), dbl(x[1]), dbl(x[2]))
    assert bad <= 20        # the library's x87 kernel rounds twice (64-bit, then 53-bit): ~2e-4 of the inputs differ by 1 ulp
