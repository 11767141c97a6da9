"""One class / resolution of tests/test_dsphmsph_pin_gpu.py with the solver's progress printed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import maxwell_amd as mx
from dsph_modes import deck_text, expected_k
lbcs = tuple(sys.argv[1].split(",")); n = int(sys.argv[2]); nev = int(sys.argv[3]); block = int(sys.argv[4])
ctx = mx.Ctx(0)
prob = mx.Problem(text=deck_text(n, lbcs, nev, basis=max(30, 4 * nev), block=block))
s = prob.solver(ctx, verbose=1)
try:
    nconv = s.solve()
    print("nconv", nconv, np.sqrt(np.sort([s.result(i)["k2"].real for i in range(nconv)])), expected_k(lbcs, nev), s.stats())
except Exception as e:
    print("FAILED:", e)
