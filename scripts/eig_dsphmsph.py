"""Config 5 physics at scale: dielectric sphere (eps 10, r 0.37) in a PEC sphere (r 0.49), octant box,
non-Hermitian Krylov-Schur + GMRES/multigrid on the GPU; wave numbers against the analytic table of the
reference's example/dsphmsph.py:512-561 (quoted below) for the symmetry class (pmc, pmc, pec)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import maxwell_amd as mx

ANALYTIC = {"TM11": 2.7914257502896397024, "TE21": 4.0593020005144315490, "TM12(21?)": 4.5810385029697019144}
ns = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 32, 64]
deck = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dsphmsph.mx")).read()
ctx = mx.Ctx(0)
out = []
for n in ns:
    text = deck.replace("resolution = [12, 12, 12]", "resolution = [%d, %d, %d]" % (n, n, n))
    t0 = time.time()
    prob = mx.Problem(text=text)
    sim = prob.sim()
    t1 = time.time()
    s = prob.solver(ctx)
    nconv = s.solve()
    st = s.stats()
    k = np.sqrt(np.sort([s.result(i)["k2"].real for i in range(nconv)]))
    rec = dict(n=n, rows=int(sim.map("bfield").size), setup_s=round(t1 - t0, 2), eigensolve_s=round(st["eigensolve_seconds"], 3),
               applies=st["applies"], vec_iters=st["vec_iters"], sca_iters=st["sca_iters"], k=[float(v) for v in k],
               rel_err_TM11=float(abs(k[0] - ANALYTIC["TM11"]) / ANALYTIC["TM11"]),
               rel_err_TE21=float(abs(k[1] - ANALYTIC["TE21"]) / ANALYTIC["TE21"]),
               max_residual=max(s.result(i)["residual"] for i in range(nconv)))
    print(json.dumps(rec), flush=True)
    out.append(rec)
    s.close(); prob.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/eig_dsphmsph.json", "w"), indent=1)
