"""K2-K9: multivector block operations through the C ABI against numpy, plus the invariant list
of Anasazi::TestMultiVecTraits that the reference's only functional test runs
(test/AnasaziInterface.cpp:53-61: complex scalars, len 5, nvecs 2)."""
import numpy as np
import pytest

import maxwell_amd as mx
from oracle import cref

pytestmark = pytest.mark.gpu

TOL = 1e-13


def _rand(rng, n, k, cx):
    a = rng.uniform(-1, 1, (n, k))
    return a + 1j * rng.uniform(-1, 1, (n, k)) if cx else a


@pytest.mark.parametrize("cx", [False, True])
@pytest.mark.parametrize("n", [5, 1000, 40003])
def test_blas1(ctx, rng, cx, n):
    k = 3
    a, b = _rand(rng, n, k, cx), _rand(rng, n, k, cx)
    A, B = mx.MultiVec.from_numpy(ctx, a), mx.MultiVec.from_numpy(ctx, b)
    assert np.array_equal(A.to_numpy(), a)
    # dot / norm2 (two-stage reduction: tolerance, not bit parity)
    ref = cref.mv_dot(a, b)
    assert np.abs(B.dot(A) - ref).max() <= TOL * n
    assert np.abs(A.norm2() - cref.mv_norm2(a)).max() <= TOL * n
    # update: this = al*A + be*this
    al, be = (0.3 - 0.2j, -1.5 + 0.1j) if cx else (0.3, -1.5)
    B.update(al, A, be)
    assert np.abs(B.to_numpy() - cref.mv_update(al, a, be, b)).max() <= 4e-16 * 4
    # update with be = 0 must not read `this` (NaN there must not propagate)
    C = mx.MultiVec.from_numpy(ctx, np.full((n, k), np.nan))
    if cx:
        C = mx.MultiVec.from_numpy(ctx, np.full((n, k), np.nan + 0j))
    C.update(al, A, 0.0)
    assert np.abs(C.to_numpy() - al * a).max() <= 1e-15
    # scale (scalar and per column), conj, fill
    A.scale(al)
    assert np.abs(A.to_numpy() - al * a).max() <= 1e-15
    cols = np.array([2.0, -1.0, 0.5]) * (1j if cx else 1.0)
    A.scale(cols)
    assert np.abs(A.to_numpy() - al * a * cols).max() <= 1e-15
    A.conj()
    assert np.abs(A.to_numpy() - np.conj(al * a * cols)).max() <= 1e-15
    A.fill(1.5 - 2j if cx else 1.5)
    assert np.all(A.to_numpy() == (1.5 - 2j if cx else 1.5))


@pytest.mark.parametrize("cx", [False, True])
@pytest.mark.parametrize("m,k", [(1, 1), (5, 1), (40, 1), (7, 2), (20, 3), (33, 8), (12, 16), (30, 4), (100, 1), (101, 5)])
def test_block_products(ctx, rng, cx, m, k):
    """K6 MvTimesMatAddMv and K7 MvTransMv for the shapes the solvers use."""
    n = 20011
    a, y = _rand(rng, n, m, cx), _rand(rng, n, k, cx)
    Bm = _rand(rng, m, k, cx)
    A, Y = mx.MultiVec.from_numpy(ctx, a), mx.MultiVec.from_numpy(ctx, y)
    al, be = (0.7 + 0.2j, -0.4 + 1.1j) if cx else (0.7, -0.4)
    got = Y.trans_mv(al, A)
    ref = cref.mv_trans_mv(al, a, y)
    assert np.abs(got - ref).max() <= TOL * n
    Y.times_mat_add_mv(al, A, Bm, be)
    ref = cref.mv_times_mat_add_mv(al, a, Bm, be, y)
    assert np.abs(Y.to_numpy() - ref).max() <= 1e-13 * m
    Y2 = mx.MultiVec.from_numpy(ctx, np.full((n, k), np.nan + 0j if cx else np.nan))
    Y2.times_mat_add_mv(1.0, A, Bm, 0.0)
    assert np.abs(Y2.to_numpy() - a @ Bm).max() <= 1e-13 * m


@pytest.mark.parametrize("cx", [False, True])
@pytest.mark.parametrize("n", [1, 64, 1001, 300007])
def test_trans_mv_with_appended_columns_and_repeated_reductions(ctx, rng, cx, n):
    """mxg_mv_trans_mv2: [A A2]^H T from one pass (the Krylov-Schur step takes V^H w and w^H w together), A a view of
    scattered columns; and dot / norm2 many times in a row: the second stage adds the partials in block order and stores
    into the pinned host buffer, every repetition returns the same bits."""
    m, k = 13, 3
    a, t = _rand(rng, n, m + 4, cx), _rand(rng, n, k, cx)
    A, T = mx.MultiVec.from_numpy(ctx, a), mx.MultiVec.from_numpy(ctx, t)
    cols = [3, 0, 16, 7, 8, 9, 10, 2, 1, 11, 12, 14, 5]
    got = T.trans_mv(1.0, A.view(cols), T)
    ref = np.conj(np.hstack([a[:, cols], t])).T @ t
    assert got.shape == (m + k, k) and np.abs(got - ref).max() <= TOL * max(n, 16)
    sep = T.trans_mv(1.0, A.view(cols))
    assert np.array_equal(got[:m], sep)                       # same kernel, same order of additions
    first_d, first_n = T.dot(T), T.norm2()
    assert np.abs(first_n - np.sqrt(np.abs(np.diag(got[m:])))).max() <= 1e-12 * max(1.0, np.sqrt(n))
    for _ in range(20):
        assert np.array_equal(T.dot(T), first_d) and np.array_equal(T.norm2(), first_n)
    u = mx.MultiVec.from_numpy(ctx, _rand(rng, n, k, cx))
    assert np.abs(u.dot(T) - cref.mv_dot(T.to_numpy(), u.to_numpy())).max() <= TOL * max(n, 16)


def test_views_clone_setblock(ctx, rng):
    n = 777
    a = rng.uniform(-1, 1, (n, 6))
    A = mx.MultiVec.from_numpy(ctx, a)
    v = A.view([5, 1, 3])                      # CloneViewNonConst, non-contiguous
    assert np.array_equal(v.to_numpy(), a[:, [5, 1, 3]])
    v.scale(2.0)                               # writes through to the parent
    a[:, [5, 1, 3]] *= 2.0
    assert np.array_equal(A.to_numpy(), a)
    c = A.clone([0, 2])                        # CloneCopy(index): deep copy
    c.fill(0.0)
    assert np.array_equal(A.to_numpy(), a)
    vv = v.view([2, 0])                        # view of a view
    assert np.array_equal(vv.to_numpy(), a[:, [3, 5]])
    b = rng.uniform(-1, 1, (n, 2))
    A.set_block(mx.MultiVec.from_numpy(ctx, b), [4, 0])   # SetBlock
    a[:, [4, 0]] = b
    assert np.array_equal(A.to_numpy(), a)
    with pytest.raises(mx.MxError, match="out of range"):
        A.view([6])


def test_add_mv_aliasing(ctx, rng):
    """MvAddMv contract this = a*A + b*B for every aliasing pattern (the reference overwrites
    `this` before reading B, src/MxAnasaziMV.cpp:108-110; the contract is what Anasazi tests)."""
    n = 501
    a, b = rng.uniform(-1, 1, (n, 2)), rng.uniform(-1, 1, (n, 2))
    A, B = mx.MultiVec.from_numpy(ctx, a), mx.MultiVec.from_numpy(ctx, b)
    C = mx.MultiVec(ctx, n, 2)
    C.add_mv(2.0, A, -3.0, B)
    assert np.abs(C.to_numpy() - (2 * a - 3 * b)).max() < 1e-15
    A.add_mv(2.0, A, -3.0, B)                  # this aliases A
    assert np.abs(A.to_numpy() - (2 * a - 3 * b)).max() < 1e-15
    B.add_mv(0.5, C, 4.0, B)                   # this aliases B
    assert np.abs(B.to_numpy() - (0.5 * (2 * a - 3 * b) + 4 * b)).max() < 1e-15


def test_random_statistics(ctx):
    """Epetra's Random() is an LCG; only the distribution is reproduced: uniform on (-1, 1)."""
    V = mx.MultiVec(ctx, 200000, 2)
    V.random(seed=42)
    v = V.to_numpy()
    assert v.min() > -1 and v.max() < 1
    assert abs(v.mean()) < 5e-3 and abs(v.var() - 1 / 3) < 5e-3
    assert abs(np.corrcoef(v[:, 0], v[:, 1])[0, 1]) < 1e-2
    W = mx.MultiVec(ctx, 200000, 2); W.random(seed=42)
    assert np.array_equal(W.to_numpy(), v)     # counter based: reproducible


def test_mask_rows(ctx, rng):
    n = 1001
    a = rng.uniform(-1, 1, (n, 2)); m = (rng.uniform(0, 1, n) > 0.3).astype(float)
    A = mx.MultiVec.from_numpy(ctx, a); M = mx.MultiVec.from_numpy(ctx, m)
    A.mask_rows(M)
    assert np.array_equal(A.to_numpy(), a * m[:, None])


def test_anasazi_multivec_traits_invariants(ctx, rng):
    """The checks of Anasazi::TestMultiVecTraits (AnasaziMVOPTester.hpp, Trilinos) on the
    reference's own test shape: MxAnasaziMV<std::complex<double>>, len 5, 2 vectors
    (test/AnasaziInterface.cpp:53-61)."""
    n, k = 5, 2
    V = mx.MultiVec(ctx, n, k, True)
    assert V.nlocal == n and V.k == k
    # MvNorm of MvInit(alpha) is |alpha| sqrt(n); MvInit(0) gives zero norms
    V.fill(0.0); assert np.all(V.norm2() == 0)
    V.fill(2.0 - 1.0j); assert np.abs(V.norm2() - abs(2 - 1j) * np.sqrt(n)).max() < 1e-14
    # MvRandom gives non-zero, distinct columns
    V.random(3); nr = V.norm2(); assert np.all(nr > 0) and nr[0] != nr[1]
    v = V.to_numpy()
    # CloneCopy is deep, CloneView is shallow
    C = V.clone(); C.fill(0.0); assert np.array_equal(V.to_numpy(), v)
    W = V.view([1]); W.fill(1.0); v[:, 1] = 1.0; assert np.array_equal(V.to_numpy(), v)
    # MvTransMv vs MvDot: diag(A^H B) == MvDot(A, B)
    B = mx.MultiVec.from_numpy(ctx, _rand(rng, n, k, True)); b = B.to_numpy()
    T = B.trans_mv(1.0, V)
    assert np.abs(np.diag(T) - B.dot(V)).max() < 1e-14
    assert np.abs(T - np.conj(v).T @ b).max() < 1e-14
    # MvTransMv with alpha
    assert np.abs(B.trans_mv(0.5j, V) - 0.5j * T).max() < 1e-14
    # MvAddMv with 0/1 coefficients
    Z = mx.MultiVec(ctx, n, k, True)
    Z.add_mv(1.0, V, 0.0, B); assert np.array_equal(Z.to_numpy(), v)
    Z.add_mv(0.0, V, 1.0, B); assert np.array_equal(Z.to_numpy(), b)
    # MvTimesMatAddMv with identity / zero and alpha, beta in {0, 1}
    I = np.eye(k, dtype=complex)
    Z.fill(3.0); Z.times_mat_add_mv(1.0, V, I, 0.0); assert np.abs(Z.to_numpy() - v).max() < 1e-15
    Z.times_mat_add_mv(0.0, V, I, 1.0); assert np.abs(Z.to_numpy() - v).max() < 1e-15
    Z.times_mat_add_mv(1.0, B, I, 1.0); assert np.abs(Z.to_numpy() - (v + b)).max() < 1e-15
    # tall-times-thin: a 2 x 1 matrix collapses to one column
    one = mx.MultiVec(ctx, n, 1, True)
    Bm = np.array([[1.0 + 1j], [2.0 - 1j]])
    one.times_mat_add_mv(1.0, V, Bm, 0.0)
    assert np.abs(one.to_numpy() - v @ Bm).max() < 1e-14
    # SetBlock then MvNorm
    Z.set_block(B.view([1]), [0]); z = Z.to_numpy(); assert np.array_equal(z[:, 0], b[:, 1])
    # MvScale
    Z.scale(0.0); assert np.all(Z.norm2() == 0)


def test_vector_storage_is_recycled_clean(ctx, rng):
    """Vector storage comes back from the context's cache (mxg_vec_alloc / release, mxg_ctx.hip): a new vector is zero
    whatever its storage held before, live vectors never share storage, and more released buffers than the cache keeps
    (32) are evicted oldest first without disturbing anybody."""
    sizes = [1000, 4096, 77777]
    live = []
    for rep in range(3):
        batch = []
        for i in range(45):                     # > the cache's buffer count
            n = sizes[i % 3] + (i // 3 if rep == 1 else 0)     # second round: many distinct sizes (forces eviction)
            v = mx.MultiVec(ctx, n, 2, is_complex=(i % 2 == 1))
            assert not np.any(v.to_numpy()), (rep, i)
            v.fill(float(rep * 100 + i + 1))
            batch.append((v, float(rep * 100 + i + 1)))
        for v, val in batch:                    # nobody wrote into anybody else's storage
            assert np.all(v.to_numpy() == val)
        for v, _ in batch[5:]:
            v.close()
        live.extend(batch[:5])
    for v, val in live:
        assert np.all(v.to_numpy() == val)
        v.close()
