"""Host prototype of the multigrid hierarchy (scipy): inner-iteration counts of the preconditioned CG for
variants of the transfer operators, against the resolution.  Experiments only; the product's cycle is
mxg_solve.hip / mx_mg.hpp.

  python scripts/mg_proto.py 32 64 96 --variant plain --variant sa
"""
import argparse
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
import maxwell_amd.api as mx  # noqa: E402


def coords(gids, ncomp, n):
    dims = (n[0] + 1, n[1] + 1, n[2] + 1)
    cell = gids // ncomp
    return gids % ncomp, cell // (dims[2] * dims[1]), (cell // dims[2]) % dims[1], cell % dims[2], dims


def aggregate(A, comp, ci, cj, ck, dims, ncomp, f=2):
    d = A.diagonal()
    active = d != 0.0
    cd = [(x + f - 1) // f for x in dims]
    key = comp + ncomp * (((ci // f) * cd[1] + (cj // f)) * cd[2] + (ck // f))
    ukeys, inv = np.unique(key[active], return_inverse=True)
    nc = len(ukeys)
    rows = np.nonzero(active)[0]
    P = sp.csr_matrix((np.ones(len(rows)), (rows, inv)), shape=(A.shape[0], nc))
    first = np.zeros(nc, dtype=np.int64)
    first[inv[::-1]] = rows[::-1]
    return P, (comp[first], ci[first] // f, cj[first] // f, ck[first] // f, cd)


def lmax_of(A, dinv, its=15, seed=0):
    v = np.random.default_rng(seed).standard_normal(A.shape[0])
    lam = 1.0
    for _ in range(its):
        v /= np.linalg.norm(v)
        v = dinv * (A @ v)
        lam = np.linalg.norm(v)
    return lam


class Level:
    pass


def build(A0, gids, n, variant, min_coarse=2000, max_levels=10, sa_omega=4.0 / 3.0, sa_levels=99, factor=2,
          filter_theta=0.0, factors=None):
    comp, ci, cj, ck, dims = coords(gids, 3, n)
    levels = []
    A = A0.tocsr()
    while True:
        L = Level()
        L.A = A
        d = A.diagonal()
        L.dinv = np.where(d != 0.0, 1.0 / np.where(d != 0.0, d, 1.0), 0.0)
        L.lmax = 1.1 * lmax_of(A, L.dinv)
        levels.append(L)
        if len(levels) >= max_levels or A.shape[0] <= min_coarse:
            break
        f = factor if len(levels) == 1 else 2
        if factors:
            f = factors[min(len(levels) - 1, len(factors) - 1)]
        if variant.startswith("sa3"):
            f = 3
        P, (comp, ci, cj, ck, dims) = aggregate(A, comp, ci, cj, ck, dims, 3, f)
        Pt = P
        if variant.startswith("sa") and len(levels) <= sa_levels:
            Af = A
            if filter_theta > 0.0:
                # drop weak connections, lump them to the diagonal
                Ac = A.tocoo()
                dd = np.abs(A.diagonal())
                keep = (np.abs(Ac.data) >= filter_theta * np.sqrt(dd[Ac.row] * dd[Ac.col])) | (Ac.row == Ac.col)
                Af = sp.csr_matrix((Ac.data[keep], (Ac.row[keep], Ac.col[keep])), shape=A.shape)
                lump = np.asarray(A.sum(axis=1)).ravel() - np.asarray(Af.sum(axis=1)).ravel()
                Af = Af + sp.diags(lump)
            dfi = L.dinv
            P = P - sp.diags(sa_omega / L.lmax * 1.1 * dfi) @ (Af @ P)
            P = P.tocsr()
        L.P = P
        L.R = P.T.tocsr()
        if variant == "sang":      # smoothed transfer, coarse operator = the tentative Galerkin operator / 1.9
            A = ((Pt.T @ A @ Pt) * (1.0 / 1.9)).tocsr()
        else:
            A = (L.R @ A @ P).tocsr()
        A.eliminate_zeros()
        if A.shape[0] == 0:
            break
    return levels


def cheb(L, r, z, steps, ratio, zero):
    return cheb_from(L, r, np.zeros_like(r) if zero else z, steps, ratio)


def cycle(levels, l, r, opts):
    L = levels[l]
    if l == len(levels) - 1:
        return cheb(L, r, None, 16, 50.0, True)
    z = cheb(L, r, None, opts["sweeps"], opts["ratio"], True)
    t = r - L.A @ z
    rc = L.R @ t
    zc = cycle(levels, l + 1, rc, opts)
    if opts.get("wcycle") and l + 1 < len(levels) - 1 and l < opts["wcycle"]:
        rc2 = rc - levels[l + 1].A @ zc
        zc = zc + cycle(levels, l + 1, rc2, opts)
    z = z + opts["cscale"] * (L.P @ zc)
    # post-smoothing: Chebyshev steps from z (restarted recurrence)
    return cheb_from(L, r, z, opts["sweeps"], opts["ratio"])


def cheb_from(L, r, z, steps, ratio):
    lmax, lmin = L.lmax, L.lmax / ratio
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma1 = theta / delta
    rho = 1.0 / sigma1
    prev = z
    for s in range(steps):
        if s == 0:
            omega, beta = 1.0 / theta, 0.0
        else:
            rn = 1.0 / (2.0 * sigma1 - rho)
            omega, beta = 2.0 * rn / delta, rn * rho
            rho = rn
        znew = z + omega * (L.dinv * (r - L.A @ z)) + beta * (z - prev)
        prev, z = z, znew
    return z


def pcg(A, b, prec, tol=1e-9, maxit=200):
    x = np.zeros_like(b)
    r = b.copy()
    z = prec(r)
    p = z.copy()
    rz = r @ z
    bn = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        q = A @ p
        a = rz / (p @ q)
        x += a * p
        r -= a * q
        if np.linalg.norm(r) <= tol * bn:
            return it
        z = prec(r)
        rz1 = r @ z
        p = z + (rz1 / rz) * p
        rz = rz1
    return maxit


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sizes", type=int, nargs="+")
    ap.add_argument("--variant", action="append")
    ap.add_argument("--cscale", type=float, default=None)
    ap.add_argument("--sweeps", type=int, default=2)
    ap.add_argument("--ratio", type=float, default=7.0)
    ap.add_argument("--sa-omega", type=float, default=4.0 / 3.0)
    ap.add_argument("--sa-levels", type=int, default=99)
    ap.add_argument("--wcycle", type=int, default=0)
    ap.add_argument("--filter", type=float, default=0.0)
    ap.add_argument("--factors", type=str, default="", help="aggregate edge per level, e.g. 2,2,4")
    args = ap.parse_args()
    shift = 0.05 * (2.0 * np.pi) ** 2
    for n in args.sizes:
        sim = mx.Sim.pillbox(n)
        A = sim.op("vecLapl", shift=shift)
        gids = sim.map("bfield")
        rng = np.random.default_rng(1)
        b = rng.standard_normal(A.shape[0]) * (A.diagonal() != 0.0)
        for v in args.variant or ["plain"]:
            t0 = time.time()
            levels = build(A, gids, [n] * 3, v, sa_omega=args.sa_omega, sa_levels=args.sa_levels, filter_theta=args.filter,
                           factors=[int(x) for x in args.factors.split(",")] if args.factors else None)
            cs = args.cscale if args.cscale is not None else (1.9 if v == "plain" else 1.0)
            opts = {"sweeps": args.sweeps, "ratio": args.ratio, "cscale": cs, "wcycle": args.wcycle}
            its = pcg(A, b, lambda r: cycle(levels, 0, r, opts))
            sizes = [(L.A.shape[0], L.A.nnz) for L in levels]
            opc = sum(s[1] for s in sizes) / sizes[0][1]
            pnnz = [levels[i].P.nnz for i in range(len(levels) - 1)]
            print("n=%d %-6s cscale %.2f: %d iterations; operator complexity %.2f; levels %s; P nnz %s (%.0f s)"
                  % (n, v, cs, its, opc, [x[0] for x in sizes], pnnz[:2], time.time() - t0), flush=True)
        sim.close()


if __name__ == "__main__":
    main()
