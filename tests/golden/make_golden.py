"""Writes the committed fixtures from the CPU oracle (run from the repo root):
    python tests/golden/make_golden.py
The reference cannot be run here (its numerics are Trilinos, not vendored), so the vectors are
outputs of oracle/yee.py, which is pinned by the analytic/known-answer tests of tests/test_oracle.py."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import yee  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
s = yee.pillbox_sim(12)
o = s.wave_ops()
L = o["vecLapl"]
np.savez_compressed(os.path.join(here, "pillbox12.npz"),
                    vecLapl_indptr=L.indptr, vecLapl_indices=L.indices, vecLapl_data=L.data,
                    dmA=o["mRhs"].diagonal(), b_gids=s.B.gids,
                    scaLapl_data=o["scaLapl"].data, curlCurl_data=o["curlCurl"].data)
json.dump(dict(n_b=len(s.B.gids), n_e=len(s.E.gids), n_psi=len(s.P.gids),
               nnz={k: int(v.nnz) for k, v in o.items()}), open(os.path.join(here, "pillbox12.json"), "w"), indent=1)
print("wrote fixtures")
