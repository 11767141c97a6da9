"""Host assembler (product, C++) against the numpy oracle: maps and every operator, bit for bit."""
import numpy as np
import pytest

import maxwell_amd as mx
from oracle import yee


def _same(a, b):
    a = a.tocsr(); b = b.tocsr()
    a.sort_indices(); b.sort_indices()
    # the product keeps explicit zeros (as Epetra does); compare on the union pattern
    d = (a - b)
    return a.shape == b.shape and (d.nnz == 0 or np.abs(d.data).max() == 0.0)


@pytest.mark.parametrize("strict", [True, False])
def test_periodic_vacuum_matches_oracle(built, strict):
    sim = mx.Sim(6, strict=strict)
    o = yee.YeeSim(6, strict=strict).wave_ops()
    for name, ref in o.items():
        assert _same(sim.op(name), ref), name


def test_pillbox_matches_oracle_bitwise(built, oracle_pillbox12):
    s, o = oracle_pillbox12
    sim = mx.Sim.pillbox(12)
    assert np.array_equal(sim.map("bfield"), s.B.gids)
    assert np.array_equal(sim.map("efield"), s.E.gids)
    assert np.array_equal(sim.map("psifield"), s.P.gids)
    for name, ref in o.items():
        got = sim.op(name)
        assert _same(got, ref), name
    L = sim.op("vecLapl"); ref = o["vecLapl"]; ref.sort_indices()
    assert np.array_equal(L.indptr, ref.indptr) and np.array_equal(L.indices, ref.indices)
    assert np.array_equal(L.data, ref.data)
    shift = yee.default_shift([1, 1, 1])
    assert _same(sim.op("vecLapl", shift=shift), yee.shifted(o["vecLapl"], o["mRhs"], shift))


def test_pec_walls_and_anisotropic_grid(built):
    """PEC/PMC box walls + non-cubic cells + an off-centre sphere (config 5 style geometry)."""
    kw = dict(origin=(0.0, 0.0, 0.0), size=(0.5, 0.6, 0.7), lower=("pmc", "pec", "pec"), upper=("pec", "pec", "pmc"))
    from oracle import geometry as geo
    o = yee.YeeSim([7, 8, 9], pec=geo.Sphere(0.49, [0.0, 0.0, 0.0]), **kw).wave_ops()
    sim = mx.Sim([7, 8, 9], pec_shape="sphere(0.49;0,0,0)", **kw)
    for name, ref in o.items():
        assert _same(sim.op(name), ref), name


def test_used_mask_marks_empty_rows(built, oracle_pillbox12):
    s, o = oracle_pillbox12
    sim = mx.Sim.pillbox(12)
    mask = sim.used_mask()
    assert np.array_equal(mask == 0.0, np.diff(o["curlE"].indptr) == 0)
    assert np.array_equal(mask == 0.0, o["mRhs"].diagonal() == 0.0)


def test_error_paths(built):
    with pytest.raises(mx.MxError, match="not found"):
        mx.Sim(4).op("nonexistent")
    with pytest.raises(mx.MxError):
        mx.Sim(4, pec_shape="blob(1;2)")
    with pytest.raises(mx.MxError, match="boundary"):
        mx.Sim(4, lower=("open", "pec", "pec"))


def test_config3_curved_pec_cylinder_matches_oracle(built):
    """BASELINE config 3 geometry at test size: z-axis PEC cylinder without caps, periodic in z
    (Dey-Mittra cut cells all along the wall: irregular nnz/row)."""
    from oracle import geometry as geo
    kw = dict(origin=(-0.5, -0.5, -0.5), size=(1.0, 1.0, 1.0))
    o = yee.YeeSim(10, pec=geo.Cylinder(0.4, [0, 0, 1], [0, 0, 0]), **kw).wave_ops()
    sim = mx.Sim(10, pec_shape="cylinder(0.4;0,0,1;0,0,0)", **kw)
    for name, ref in o.items():
        assert _same(sim.op(name), ref), name
    rows = np.diff(o["vecLapl"].indptr)
    assert rows.max() > 7 and (rows == 7).sum() > 0      # wall rows wider than the bulk stencil
