"""N > 1 path on CPU (gloo): partition rules, box assembly with halo, halo plan and exchange."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("case,world", [("pillbox", 2), ("periodic", 2), ("pillbox", 4), ("pillbox", 8)])
def test_box_assembly_and_halo_plan_gloo(built, case, world):
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), case]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "DIST_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_partition_follows_reference_rules(built):
    """MxGrid::partition: first minimum of sum p_i/n_i over the enumeration order of
    unorderedFactorizations x permutations (src/MxGrid.cpp:100-165); boxes of getGridDomain (:171-193)."""
    from maxwell_amd import distributed as mxd
    assert mxd.partition([256] * 3, 1, 0) == ([1, 1, 1], [0, 0, 0], [257, 257, 257])
    assert mxd.partition([256] * 3, 2, 1) == ([2, 1, 1], [128, 0, 0], [257, 257, 257])
    assert mxd.partition([256] * 3, 4, 3)[0] == [2, 2, 1]
    assert mxd.partition([512] * 3, 8, 7) == ([2, 2, 2], [256, 256, 256], [513, 513, 513])
    # rank -> box: last dimension fastest
    assert mxd.partition([512] * 3, 8, 1)[1] == [0, 0, 256]
    # anisotropic grid: the long axis gets the processes (2 x N x N fibre grid of example/cylinder.py)
    assert mxd.partition([2, 240, 240], 4, 0)[0] == [1, 2, 2]
    # 6 ranks on a 2:3:1 box
    pg = mxd.partition([64, 96, 32], 6, 0)[0]
    assert sorted(pg) == [1, 2, 3] and pg[1] == 3
    # uneven split: floor(n/p + 0.5) cells per box, the last box takes the rest + the node layer
    assert mxd.partition([10, 10, 10], 3, 2)[1:] == ([6, 0, 0], [11, 11, 11])


def test_owner_of_gids_matches_boxes(built):
    from maxwell_amd import distributed as mxd
    n = [10, 7, 9]
    world = 6
    pg = mxd.partition(n, world, 0)[0]
    ni = [v + 1 for v in n]
    cells = np.stack(np.meshgrid(*[np.arange(v) for v in ni], indexing="ij"), -1).reshape(-1, 3)
    cellidx = (cells[:, 0] * ni[1] + cells[:, 1]) * ni[2] + cells[:, 2]
    owner = mxd.owner_of_gids(cellidx * 3 + 1, 3, n, pg)
    for r in range(world):
        _, lo, hi = mxd.partition(n, world, r)
        inside = np.all((cells >= lo) & (cells < hi), axis=1)
        assert np.array_equal(owner == r, inside)


@pytest.mark.parametrize("world", [2, 4])
def test_box_assembly_with_dielectric_is_exact(built, world):
    """Config 5 geometry (eps = 10 sphere in a PEC sphere, octant box) on MxGridDomain boxes: the FIT inverse
    permittivity reaches one cell further than the vacuum stencils, the box + two-cell halo window still gives
    every rank exactly its rows of the global operators (values bit for bit) - no GPU, no communication."""
    import scipy.sparse as sp
    import maxwell_amd as mx
    from maxwell_amd import distributed as mxd
    kw = dict(origin=(0.0,) * 3, size=(0.5,) * 3, lower=("pmc", "pmc", "pec"), upper=("pec",) * 3,
              pec_shape="sphere(0.49;0,0,0)", dielectrics=[(10.0, "sphere(0.37;0,0,0)")])
    n = [12, 10, 12]
    g = mx.Sim(n, **kw)
    gB, gP, gE = g.map("bfield"), g.map("psifield"), g.map("efield")
    fields = {"shifted": (gB, gB), "scaLapl": (gP, gP), "divB": (gP, gB), "gradPsi": (gB, gP), "curlCurl": (gB, gB),
              "curlB": (gE, gB)}
    seen = {k: 0 for k in fields}
    for rank in range(world):
        b = mxd.BoxSim(n, rank, world, **kw)
        for name, (rf, cf) in fields.items():
            G = (g.op("vecLapl", shift=0.5) if name == "shifted" else g.op(name)).tocsr()
            d = b.dist_op(name, 0.5)
            rows = np.searchsorted(rf, d["rowgid"])
            assert np.array_equal(rf[rows], d["rowgid"])
            cols = np.searchsorted(cf, d["colgid"])
            assert np.array_equal(cf[cols], d["colgid"])
            M = sp.csr_matrix((d["vals"], cols, d["rowptr"]), shape=(len(rows), len(cf)))
            D = M - G[rows]
            assert D.nnz == 0 or np.abs(D.data).max() == 0.0, (name, rank)
            seen[name] += len(rows)
    assert seen["shifted"] == len(gB) and seen["scaLapl"] == len(gP) and seen["curlB"] == len(gE)
