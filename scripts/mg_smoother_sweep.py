"""Eigen-solve of the pillbox at --cells^3 under the multigrid smoother variants (one subprocess per variant, the
smoother goes in as a solver parameter, the Chebyshev ratio and the coarse-correction scale as the experiment
switches MXG_MG_CHEB_RATIO / MXG_MG_CSCALE read when the hierarchy is created): damped Jacobi vs Chebyshev with several
eigenvalue ratios and coarse-correction scales.  Prints inner iterations and solve time per variant."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, sys, time
import numpy as np
sys.path.insert(0, %r)
import maxwell_amd as mx
n = int(sys.argv[1]); smoother = sys.argv[2]; sweeps = int(sys.argv[3])
ctx = mx.Ctx(0)
sim = mx.Sim.pillbox(n)
p = mx.SolverParams(nev=5, basis=30, tol=1e-8, shift=0.05 * (2.0 * np.pi) ** 2, lin_type="cg", lin_tol=1e-9, prec_type="amg", amg_smoother=smoother, amg_sweeps=sweeps)
s = mx.Solver(ctx, sim, p)
nconv = s.solve()
st = s.stats()
print("RESULT " + json.dumps(dict(conv=nconv, vec=st["vec_iters"], sca=st["sca_iters"], secs=st["eigensolve_seconds"],
                                  k2=[s.result(i)["k2"].real for i in range(nconv)])))
''' % ROOT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=128)
    ap.add_argument("--out", default=None)
    ap.add_argument("--ratios", default="8,14,20,30")
    ap.add_argument("--cscales", default="1.0,1.5,1.9")
    ap.add_argument("--sweeps", type=int, default=2, help="smoother steps per level (the deck's `sweeps`)")
    ap.add_argument("--coarse", default="", help="coarsest-level Chebyshev variants, e.g. 10:30,10:100 (degree:ratio)")
    a = ap.parse_args()
    variants = [("jacobi", None, None)]
    for ratio in [float(v) for v in a.ratios.split(",")]:
        for cs in [float(v) for v in a.cscales.split(",")]:
            variants.append(("cheb", ratio, cs))
    variants = [v + (None,) for v in variants]
    for cc in [c for c in a.coarse.split(",") if c]:
        variants.append(("cheb", float(a.ratios.split(",")[0]), float(a.cscales.split(",")[-1]), cc))
    rows = []
    for sm, ratio, cs, cc in variants:
        env = dict(os.environ)
        if cc: env["MXG_MG_COARSE_CHEB"] = cc
        if ratio: env["MXG_MG_CHEB_RATIO"] = str(ratio)
        if cs: env["MXG_MG_CSCALE"] = str(cs)
        r = subprocess.run([sys.executable, "-c", WORKER, str(a.cells), "Jacobi" if sm == "jacobi" else "Chebyshev", str(a.sweeps)], capture_output=True, text=True, env=env, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        res = json.loads(line[0][7:]) if line else {"error": r.stderr[-300:]}
        res.update(smoother=sm, ratio=ratio, cscale=cs, coarse=cc, sweeps=a.sweeps)
        rows.append(res)
        print(json.dumps(res), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
