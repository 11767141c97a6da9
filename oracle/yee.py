"""Oracle: Yee / Dey-Mittra operator assembly and the MxMagWaveOp matrices
(TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

numpy/scipy restatement, 3-D only, of

  lattice / indices   src/MxGrid.h:96-128, src/MxGrid.cpp:171-193 (single rank:
                      lb = 0, ub = N+1), src/MxGridDomain.cpp:21-89
  field layouts       src/MxYeeMagFieldBase.cpp:85-116, src/MxYeeElecFieldBase.cpp:64-82,
                      src/MxYeePsiField.cpp:50-70
  BC translation      src/MxYeeMagFieldBase.cpp:119-168, src/MxYeeElecFieldBase.cpp:89-138,
                      src/MxYeePsiField.cpp:73-96
  map membership      src/MxGridField.cpp:41-77,256-297, src/MxYeeFitBField.cpp:63-80,
                      src/MxYeeFitEField.cpp:53-88, src/MxYeePsiField.cpp:100-114
  BC mirroring/phase  src/MxGridField.cpp:80-190 (incl. the two-'if' quirk at :101-116),
                      src/MxYeeFitBField.cpp:83-90, src/MxYeeFitEField.cpp:91-98,
                      src/MxYeePsiField.cpp:117-124
  fractions           src/MxGridField.hpp:177-183, src/MxGridField.cpp:193-226
  operators           src/MxYeeDeyMittraCurlE.cpp:117-179, CurlB.cpp:113-169,
                      DivB.cpp:157-200, GradPsi.cpp:23-84, Fracs.cpp:30-131
                      (minFrac arguments of src/MxEMOps.cpp:56-63,125-127 are all 0)
  products            src/MxMagWaveOp.cpp:138-273, src/MxEMOps.h:61-74,
                      src/MxUtil.cpp:318-371, src/MxCrsMatrix.cpp:84-116 (purgeZeros)

Reference quirk kept in `strict` mode (the default): MxYeeFitEField::getCompFactor
and MxYeePsiField::getCompFactor test useCompInMap on the *un-interiorised* cell
(src/MxYeeFitEField.cpp:96, src/MxYeePsiField.cpp:122), so on a PERIODIC axis the
wrap-around E entries of curlE at the upper boundary are dropped
(src/MxGridField.cpp:70-73 returns false for cell == N, xi == 0).  A PEC cavity that
does not touch the domain boundary (the pillbox) is unaffected.  `strict=False`
evaluates membership on the interiorised cell instead, which restores
curlE = curlB^T on periodic boxes and is what the periodic known-answer tests use.
"""
import numpy as np
import scipy.sparse as sp

from . import geometry as geo

PERIODIC, ZERO, CONSTANT = 0, 1, 2


def _bc(name):
    return {"periodic": "periodic", "pec": "pec", "pmc": "pmc"}[name.lower()]


class _Field:
    def __init__(self, name, ncomp, coords, dirs):
        self.name = name
        self.ncomp = ncomp
        self.coords = coords      # (ncomp, 3) cell-relative offsets in units of the cell size
        self.dirs = dirs          # component direction (None for psi)
        self.lbc = np.zeros((ncomp, 3), dtype=int)
        self.ubc = np.zeros((ncomp, 3), dtype=int)
        self.frac = None          # region ('pec') fractions: list over comps of ext-lattice arrays, or None
        self.reps = {}            # other shape reps by name (dielectrics): same layout, not a region
        self.gids = None          # sorted global ids in the map
        self.comp = None
        self.cells = None


class Pml:
    """MxPML, src/MxPML.h:31-175: a stretched-coordinate layer whose stretch s = 1 + i alpha(x) does not depend on
    frequency (alpha = alphaMax ((x - begin) . dir / width)^exponent), so the eigenproblem stays linear.  getInvS /
    getS return dir dir^T s + (I - dir dir^T) and dir dir^T / s + (I - dir dir^T).  isDiag() is false for every
    layer: setIsDiag counts |dir_i| < MxUtil::dEps, and dEps is 0.0 (src/MxUtil.hpp:26), so no component is ever
    counted as zero - every row inside a layer takes the 9-point stencil path.  Complex arithmetic as in the rest of
    this file: native (naive) products, the explicit naive quotient YeeSim._cdiv."""

    def __init__(self, shape, begin, end, alpha_max=1.0, exponent=2.0, name="PML0"):
        self.shape, self.name = shape, name
        self.alpha_max, self.exponent = float(alpha_max), float(exponent)
        self.start = [float(v) for v in begin]
        self.end = [float(v) for v in end]
        d = [self.end[i] - self.start[i] for i in range(3)]
        import math
        self.width = math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])          # :110-112
        self.dir = [d[i] / self.width for i in range(3)]
        dc = [complex(v, 0.0) for v in self.dir]
        self.outer = [[dc[i] * dc[j] for j in range(3)] for i in range(3)]       # setOuterProds, :85-92
        self.compl = [[complex(1.0 if i == j else 0.0, 0.0) - self.outer[i][j] for j in range(3)] for i in range(3)]
        self.is_diag = False                                                      # setIsDiag with dEps = 0, :95-105

    def alpha(self, point):                                                       # getAlpha, :168-171
        t = ((point[0] - self.start[0]) * self.dir[0] + (point[1] - self.start[1]) * self.dir[1] +
             (point[2] - self.start[2]) * self.dir[2]) / self.width
        return self.alpha_max * float(np.float64(t) ** np.float64(self.exponent))

    def inv_s(self, point):                                                       # getInvS, :145-149
        s = complex(1.0, self.alpha(point))
        return [[s * self.outer[i][j] + self.compl[i][j] for j in range(3)] for i in range(3)]

    def s(self, point):                                                           # getS, :151-161 (m / s = m * (1 / s))
        r = YeeSim._cdiv(complex(1.0, 0.0), complex(1.0, self.alpha(point)))
        return [[r * self.outer[i][j] + self.compl[i][j] for j in range(3)] for i in range(3)]


class YeeSim:
    """Single-rank 3-D restatement of MxEMSim::setup + MxEMOps + MxMagWaveOp::initMatrices."""

    def __init__(self, N, origin=(0., 0., 0.), size=(1., 1., 1.),
                 lower=("periodic",) * 3, upper=("periodic",) * 3,
                 pec=None, phase_shifts=(0., 0., 0.), strict=True, dm_frac=0.0, dielectrics=(), mus=(), pmls=()):
        self.n = np.array([int(N)] * 3 if np.isscalar(N) else [int(x) for x in N])
        self.origin = np.array(origin, dtype=float)
        self.size = np.array(size, dtype=float)
        self.d = self.size / self.n.astype(float)          # src/MxGrid.cpp:18
        self.pec = pec
        self.strict = strict
        self.dm_frac = float(dm_frac)   # <PEC> 'D-M frac', src/MxProblem.cpp:450
        self.phase = np.exp(1j * np.array(phase_shifts, dtype=float))   # src/MxGridField.cpp:34-38
        self.is_complex = bool(np.any(np.array(phase_shifts) != 0.0))   # or lossy dielectrics, set below
        lower = [_bc(x) for x in lower]
        upper = [_bc(x) for x in upper]

        half = 0.5
        B = _Field("bfield", 3, np.array([[0, half, half], [half, 0, half], [half, half, 0]]), [0, 1, 2])
        E = _Field("efield", 3, np.array([[half, 0, 0], [0, half, 0], [0, 0, half]]), [0, 1, 2])
        P = _Field("psifield", 1, np.array([[half, half, half]]), [None])
        self.B, self.E, self.P = B, E, P
        # dielectrics: (shape, 3x3 real symmetric eps, name), src/MxProblem.cpp:487-527 / MxDielectric.cpp
        self.dielectrics = []
        for sh, eps, name in dielectrics:
            e = np.array(eps).reshape(3, 3)
            # a loss tangent makes eps complex (src/MxProblem.cpp:505-513): keep floats otherwise
            e = e.astype(complex) if np.iscomplexobj(e) and np.any(e.imag != 0.0) else e.real.astype(float)
            self.dielectrics.append((sh, e, name))
        # perfectly matched layers (Pml objects): complex materials; the D and H fields and both material operators are
        # then built whether or not there is a dielectric / mu object (src/MxEMSim.cpp:89,96, src/MxEMOps.cpp:66,131)
        self.pmls = list(pmls)
        self.lossy = any(np.iscomplexobj(e) for _, e, _ in self.dielectrics) or bool(self.pmls)
        self.is_complex = self.is_complex or self.lossy
        # D lives where E lives (MxYeeElecFieldBase) but is represented by the dual face
        # (src/MxYeeFitDField.cpp:42-52); only built with dielectrics (src/MxEMSim.cpp:83-88)
        D = _Field("dfield", 3, E.coords.copy(), [0, 1, 2]) if (self.dielectrics or self.pmls) else None
        self.D = D
        # magnetic materials: (shape, 3x3 real symmetric mu, name), src/MxProblem.cpp:515-556 / MxMu.cpp.  H lives where B
        # lives (MxYeeMagFieldBase) but is represented by the dual edge through the face centre (src/MxYeeFitHField.cpp:37-47);
        # only built with mu objects (src/MxEMSim.cpp:95-100)
        self.mus = []
        for sh, mu, name in mus:
            m = np.array(mu).reshape(3, 3)
            # a loss tangent makes the diagonal complex, mu_ii (1 + i t) (src/MxProblem.cpp:536-549): keep floats otherwise
            m = m.astype(complex) if np.iscomplexobj(m) and np.any(m.imag != 0.0) else m.real.astype(float)
            self.mus.append((sh, m, name))
        self.mu_lossy = any(np.iscomplexobj(m) for _, m, _ in self.mus)
        self.is_complex = self.is_complex or self.mu_lossy
        H = _Field("hfield", 3, B.coords.copy(), [0, 1, 2]) if (self.mus or self.pmls) else None
        self.H = H
        for F, normal_pec, tang_pec in ((B, ZERO, CONSTANT), (E, CONSTANT, ZERO)) + (((D, CONSTANT, ZERO),) if D else ()) + \
                (((H, ZERO, CONSTANT),) if H else ()):
            for c in range(3):
                for i in range(3):
                    for bcs, arr in ((lower, F.lbc), (upper, F.ubc)):
                        if bcs[i] == "pec":
                            arr[c, i] = normal_pec if F.dirs[c] == i else tang_pec
                        elif bcs[i] == "pmc":
                            arr[c, i] = tang_pec if F.dirs[c] == i else normal_pec
                        else:
                            arr[c, i] = PERIODIC
        for i in range(3):
            for bcs, arr in ((lower, P.lbc), (upper, P.ubc)):
                arr[0, i] = {"pec": CONSTANT, "pmc": ZERO, "periodic": PERIODIC}[bcs[i]]

        fields = [F for F in (B, E, D, H, P) if F is not None]
        if pec is not None:
            for F in fields:
                F.frac = self._add_shape_rep(F, pec)            # src/MxEMSim.cpp:121-129 (region)
        for sh, _, name in self.dielectrics:                    # src/MxEMSim.cpp:131-146 (not on B)
            for F in fields:
                if F is not B and F is not H:
                    F.reps[name] = self._add_shape_rep(F, sh)
        for sh, _, name in self.mus:                            # src/MxEMSim.cpp:148-163 (not on E, D)
            for F in fields:
                if F is not E and F is not D:
                    F.reps[name] = self._add_shape_rep(F, sh)
        for F in (B, E, P):
            self._set_map(F)
        if D is not None:                                       # dfield->setMap(efield), src/MxEMSim.cpp:196-197
            D.gids, D.comp, D.cells = E.gids, E.comp, E.cells
        if H is not None:                                       # hfield->setMap(bfield), src/MxEMSim.cpp:198-199
            H.gids, H.comp, H.cells = B.gids, B.comp, B.cells

    # ---------------------------------------------------------------- lattice helpers
    def node_coord(self, cells):
        return self.origin + cells.astype(float) * self.d           # src/MxGrid.cpp:266-271

    def cell_to_global(self, cells):
        """src/MxGrid.h:98-113 (C order over the (N+1)^3 node lattice, wrap with N+1)."""
        ni = self.n + 1
        c = np.where(cells >= ni, cells - ni, np.where(cells < 0, ni + cells, cells))
        return (c[..., 0] * ni[1] + c[..., 1]) * ni[2] + c[..., 2]

    def global_to_cell(self, idx):
        ni = self.n + 1
        k = idx % ni[2]
        j = (idx // ni[2]) % ni[1]
        i = idx // (ni[2] * ni[1])
        return np.stack([i, j, k], axis=-1)

    def interior(self, F, comp, cells):
        """getInteriorComp, src/MxGridField.cpp:80-142 (case 1 is a stand-alone `if`,
        so case 2 overrides it for cell < 0)."""
        new = cells.copy()
        for i in range(3):
            lbc, ubc = F.lbc[comp, i], F.ubc[comp, i]
            xi0 = F.coords[comp, i] == 0.0
            ci = cells[..., i]
            ni = self.n[i]
            out = ci.copy()
            below = ci < 0
            if lbc == PERIODIC:
                out = np.where(below, ni + ci, out)
            else:
                out = np.where(below, -ci - 1, out)   # case 2 wins over case 1
            if xi0:
                on_ub = ci == ni
                above = ci > ni
                if ubc == PERIODIC:
                    out = np.where(on_ub, 0, out)
                    out = np.where(above, ci - ni, out)
                else:
                    out = np.where(above, ni - (ci - ni), out)
            else:
                above = ci >= ni
                if ubc == PERIODIC:
                    out = np.where(above, ci - ni, out)
                else:
                    out = np.where(above, ni - (ci - ni + 1), out)
            new[..., i] = out
        return new

    def comp_coord(self, F, comp, cells):
        return self.node_coord(cells) + F.coords[comp] * self.d

    def frac_at(self, F, comp, cells, name=None):
        """getCompFrac, src/MxGridField.hpp:258-277: one guard cell, 1.0 without that shape rep."""
        rep = F.frac if name is None else F.reps.get(name)
        if rep is None:
            return np.ones(cells.shape[:-1])
        return rep[comp][cells[..., 0] + 1, cells[..., 1] + 1, cells[..., 2] + 1]

    def _ext_cells(self):
        ax = [np.arange(-1, self.n[i] + 2) for i in range(3)]
        g = np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1)
        return g.reshape(-1, 3)

    def _add_shape_rep(self, F, shape):
        """addShapeRep with one guard cell, src/MxGridField.cpp:193-226."""
        cells = self._ext_cells()
        ext_shape = tuple(self.n + 3)
        out = []
        for comp in range(F.ncomp):
            pos = self.comp_coord(F, comp, self.interior(F, comp, cells))
            if F is self.B or F is self.D:      # face / dual face normal to the component
                c1, c2 = (comp + 1) % 3, (comp + 2) % 3
                fr = geo.rect_fractions(shape, pos, comp, self.d[c1], self.d[c2])
            elif F is self.E or (self.H is not None and F is self.H):
                fr = geo.segment_fractions(shape, pos, comp, self.d[comp])
            else:
                fr = geo.box_fractions(shape, pos, self.d[0], self.d[1], self.d[2])
            out.append(fr.reshape(ext_shape))
        return out

    # ---------------------------------------------------------------- membership / factors
    def base_use(self, F, comp, cells):
        """MxGridField::useCompInMap, src/MxGridField.cpp:41-77."""
        ok = np.ones(cells.shape[:-1], dtype=bool)
        if F.frac is not None:
            ok &= self.frac_at(F, comp, cells) != 0.0
        for i in range(3):
            ci = cells[..., i]
            xi0 = F.coords[comp, i] == 0.0
            ni = self.n[i]
            if xi0:
                if F.lbc[comp, i] == ZERO:
                    ok &= ~(ci == 0)
                if F.ubc[comp, i] in (ZERO, PERIODIC):
                    ok &= ~(ci == ni)
            else:
                ok &= ~(ci == ni)
        return ok

    def b_use(self, cells):
        """MxYeeFitBField::useCompInMap, src/MxYeeFitBField.cpp:63-80: any used
        component keeps the whole cell (the dmFrac test after the loop is unreachable: when the
        loop falls through, the base rule is already false for `comp`)."""
        u = self.base_use(self.B, 0, cells)
        for c in (1, 2):
            u = u | self.base_use(self.B, c, cells)
        return u

    def e_use(self, comp, cells):
        """MxYeeFitEField::useCompInMap, src/MxYeeFitEField.cpp:53-88."""
        u = self.base_use(self.E, comp, cells)
        c2, c3 = (comp + 1) % 3, (comp + 2) % 3
        m3 = cells.copy(); m3[..., c3] -= 1
        m2 = cells.copy(); m2[..., c2] -= 1
        return u & self.b_use(cells) & self.b_use(m3) & self.b_use(m2)

    def psi_use(self, cells):
        """MxYeePsiField::useCompInMap, src/MxYeePsiField.cpp:100-114."""
        u = self.base_use(self.P, 0, cells)
        anyb = self.b_use(cells)
        for i in range(3):
            p = cells.copy(); p[..., i] += 1
            anyb = anyb | self.b_use(p)
        return u & anyb

    def use(self, F, comp, cells):
        if F is self.B:
            return self.b_use(cells)
        if F is self.E:
            return self.e_use(comp, cells)
        return self.psi_use(cells)

    def base_factor(self, F, comp, cells):
        """MxGridField::getCompFactor, src/MxGridField.cpp:145-190."""
        res = np.ones(cells.shape[:-1], dtype=complex)
        if F.frac is not None:
            res = np.where(self.frac_at(F, comp, cells) == 0.0, 0.0, res)
        for i in range(3):
            ci = cells[..., i]
            lbc, ubc = F.lbc[comp, i], F.ubc[comp, i]
            xi0 = F.coords[comp, i] == 0.0
            ni = self.n[i]
            below = ci < 0
            if lbc == PERIODIC:
                res = np.where(below, res / self.phase[i], res)
            elif lbc == ZERO:
                res = np.where(below, -res, res)
            if xi0:
                if lbc == ZERO:
                    res = np.where(ci == 0, 0.0, res)
                on_ub = ci == ni
                if ubc == PERIODIC:
                    res = np.where(on_ub, res * self.phase[i], res)
                elif ubc == ZERO:
                    res = np.where(on_ub, 0.0, res)
                above = ci > ni
            else:
                above = ci >= ni
            if ubc == PERIODIC:
                res = np.where(above, res * self.phase[i], res)
            elif ubc == ZERO:
                res = np.where(above, -res, res)
        return res

    def factor(self, F, comp, cells):
        res = self.base_factor(F, comp, cells)
        if F is self.B:
            # MxYeeFitBField::getCompFactor, src/MxYeeFitBField.cpp:83-90
            if F.frac is not None and self.dm_frac > 0.0:
                res = np.where(self.frac_at(F, comp, cells) < self.dm_frac, 0.0, res)
            return res
        if F is self.D:                       # no override: MxGridField::getCompFactor
            return res
        probe = cells if self.strict else self.interior(F, comp, cells)
        return np.where(self.use(F, comp, probe), res, 0.0)

    def gid(self, F, comp, cells):
        """globCompIndx, src/MxGridField.hpp:142-145."""
        return comp + F.ncomp * self.cell_to_global(self.interior(F, comp, cells))

    # ---------------------------------------------------------------- maps
    def _set_map(self, F):
        """setMap, src/MxGridField.cpp:256-297: interior cells in C order, comps inner."""
        ax = [np.arange(0, self.n[i] + 1) for i in range(3)]
        cells = np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1).reshape(-1, 3)
        keep = np.stack([self.use(F, c, cells) for c in range(F.ncomp)], axis=1)   # (ncell, ncomp)
        cell_idx, comp = np.nonzero(keep)
        F.comp = comp
        F.cells = cells[cell_idx]
        F.gids = comp + F.ncomp * self.cell_to_global(F.cells)
        assert np.all(np.diff(F.gids) > 0)

    def lid(self, F, gids):
        pos = np.searchsorted(F.gids, gids)
        pos = np.clip(pos, 0, len(F.gids) - 1)
        ok = F.gids[pos] == gids
        return pos, ok

    # ---------------------------------------------------------------- operators
    def _assemble(self, rowF, colF, rows, entries):
        """entries: list of (col_comp, col_cells, value arrays).  Zero factors are
        dropped before insertion exactly like the reference's `if (factors[i] != 0)`."""
        r_all, c_all, v_all = [], [], []
        for row_lid, col_comp, col_cells, vals in entries:
            fac = self.factor(colF, col_comp, col_cells)
            nz = fac != 0
            g = self.gid(colF, col_comp, col_cells[nz])
            pos, ok = self.lid(colF, g)
            if not np.all(ok):
                raise AssertionError("column outside the domain map in %s x %s" % (rowF.name, colF.name))
            r_all.append(row_lid[nz]); c_all.append(pos); v_all.append((fac[nz] * vals))
        r = np.concatenate(r_all); c = np.concatenate(c_all); v = np.concatenate(v_all)
        if not self.is_complex:
            v = v.real
        m = sp.coo_matrix((v, (r, c)), shape=(len(rowF.gids), len(colF.gids))).tocsr()
        m.sum_duplicates()
        m.sort_indices()
        return m

    def curlE(self):
        """src/MxYeeDeyMittraCurlE.cpp:117-179 (rows: B map, cols: E map)."""
        inv = 1.0 / self.d
        entries = []
        for c0 in range(3):
            c1, c2 = (c0 + 1) % 3, (c0 + 2) % 3
            sel = np.nonzero(self.B.comp == c0)[0]
            cells = self.B.cells[sel]
            live = self.factor(self.B, c0, cells) != 0          # :148-150
            sel, cells = sel[live], cells[live]
            p1 = cells.copy(); p1[:, c1] += 1
            p2 = cells.copy(); p2[:, c2] += 1
            entries += [(sel, c2, p1, inv[c1]), (sel, c2, cells, -inv[c1]),
                        (sel, c1, p2, -inv[c2]), (sel, c1, cells, inv[c2])]
        return self._assemble(self.B, self.E, None, entries)

    def curlB(self):
        """src/MxYeeDeyMittraCurlB.cpp:113-169 (rows: E map, cols: B map)."""
        inv = 1.0 / self.d
        entries = []
        for c0 in range(3):
            c1, c2 = (c0 + 1) % 3, (c0 + 2) % 3
            sel = np.nonzero(self.E.comp == c0)[0]
            cells = self.E.cells[sel]
            m1 = cells.copy(); m1[:, c1] -= 1
            m2 = cells.copy(); m2[:, c2] -= 1
            entries += [(sel, c2, cells, inv[c1]), (sel, c2, m1, -inv[c1]),
                        (sel, c1, cells, -inv[c2]), (sel, c1, m2, inv[c2])]
        return self._assemble(self.E, self.B, None, entries)

    def divB(self):
        """src/MxYeeDeyMittraDivB.cpp:157-200 (rows: Psi map, cols: B map)."""
        inv = 1.0 / self.d
        sel = np.arange(len(self.P.gids))
        cells = self.P.cells
        entries = []
        for c in range(3):
            p = cells.copy(); p[:, c] += 1
            entries += [(sel, c, p, inv[c]), (sel, c, cells, -inv[c])]
        return self._assemble(self.P, self.B, None, entries)

    def gradPsi(self):
        """src/MxYeeDeyMittraGradPsi.cpp:23-84 (rows: B map, cols: Psi map)."""
        inv = 1.0 / self.d
        entries = []
        for c in range(3):
            sel = np.nonzero(self.B.comp == c)[0]
            cells = self.B.cells[sel]
            live = self.factor(self.B, c, cells) != 0           # :58-60
            sel, cells = sel[live], cells[live]
            m = cells.copy(); m[:, c] -= 1
            entries += [(sel, 0, cells, inv[c]), (sel, 0, m, -inv[c])]
        return self._assemble(self.B, self.P, None, entries)

    def fracs(self, F, inverse=False):
        """src/MxYeeDeyMittraFracs.cpp:30-131 with minFrac = 0: diagonal of the map's
        fractions, or their reciprocals (0 stays 0)."""
        v = self._frac_vec(F)
        if inverse:
            v = np.where(v == 0.0, 0.0, 1.0 / np.where(v == 0.0, 1.0, v))
        if self.is_complex:
            v = v.astype(complex)
        return sp.diags(v, format="csr")

    def _frac_vec(self, F):
        out = np.empty(len(F.gids))
        for c in range(F.ncomp):
            s = F.comp == c
            out[s] = self.frac_at(F, c, F.cells[s])
        return out


    # ---------------------------------------------------------------- FIT inverse permittivity
    @staticmethod
    def _eps_is_diag(eps):
        """MxDielectric::checkIsDiag, src/MxDielectric.cpp:25-34 (MxUtil::dEps = 0: any non-zero upper-triangle entry)."""
        return not (abs(eps[0, 1]) > 0.0 or abs(eps[0, 2]) > 0.0 or abs(eps[1, 2]) > 0.0)

    @staticmethod
    def _gamma_pi(eps, n):
        """gamma = I + n n^T (I - eps) / (n . eps n), pi = eps gamma (src/MxYeeFitInvEps.cpp:345-347,365-366);
        plain float loops in a fixed order so that the host code can match bit for bit."""
        epsn = [eps[i][0] * n[0] + eps[i][1] * n[1] + eps[i][2] * n[2] for i in range(3)]
        denom = n[0] * epsn[0] + n[1] * epsn[1] + n[2] * epsn[2]
        ime = [[(1.0 if i == j else 0.0) - eps[i][j] for j in range(3)] for i in range(3)]
        nn = [[n[i] * n[j] for j in range(3)] for i in range(3)]
        prod = geo._matmul3(nn, ime)
        gamma = [[(1.0 if i == j else 0.0) + prod[i][j] / denom for j in range(3)] for i in range(3)]
        pi = geo._matmul3(eps, gamma)
        return gamma, pi

    # complex permittivity: the same formulas with explicit complex division (the naive quotient; language
    # runtimes differ in how they scale it), products and sums as the native (naive) complex operations
    @staticmethod
    def _cdiv(a, b):
        a = complex(a); b = complex(b)
        den = b.real * b.real + b.imag * b.imag
        return complex((a.real * b.real + a.imag * b.imag) / den, (a.imag * b.real - a.real * b.imag) / den)

    @classmethod
    def _matmul3c(cls, x, y):
        out = [[0j] * 3 for _ in range(3)]
        for i in range(3):
            for j in range(3):
                acc = 0j
                for k in range(3):
                    acc = acc + complex(x[i][k]) * complex(y[k][j])
                out[i][j] = acc
        return out

    @classmethod
    def _inverse3c(cls, a):
        a = [[complex(v) for v in row] for row in a]
        det = (a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) -
               a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
               a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]))
        d = cls._cdiv
        return [[d(a[1][1] * a[2][2] - a[1][2] * a[2][1], det), d(a[0][2] * a[2][1] - a[0][1] * a[2][2], det),
                 d(a[0][1] * a[1][2] - a[0][2] * a[1][1], det)],
                [d(a[1][2] * a[2][0] - a[1][0] * a[2][2], det), d(a[0][0] * a[2][2] - a[0][2] * a[2][0], det),
                 d(a[0][2] * a[1][0] - a[0][0] * a[1][2], det)],
                [d(a[1][0] * a[2][1] - a[1][1] * a[2][0], det), d(a[0][1] * a[2][0] - a[0][0] * a[2][1], det),
                 d(a[0][0] * a[1][1] - a[0][1] * a[1][0], det)]]

    @classmethod
    def _gamma_pi_c(cls, eps, n):
        eps = [[complex(v) for v in row] for row in eps]
        nc = [complex(v, 0.0) for v in n]
        epsn = [eps[i][0] * nc[0] + eps[i][1] * nc[1] + eps[i][2] * nc[2] for i in range(3)]
        denom = nc[0] * epsn[0] + nc[1] * epsn[1] + nc[2] * epsn[2]
        ime = [[complex(1.0 if i == j else 0.0, 0.0) - eps[i][j] for j in range(3)] for i in range(3)]
        nn = [[nc[i] * nc[j] for j in range(3)] for i in range(3)]
        prod = cls._matmul3c(nn, ime)
        gamma = [[complex(1.0 if i == j else 0.0, 0.0) + cls._cdiv(prod[i][j], denom) for j in range(3)] for i in range(3)]
        pi = cls._matmul3c(eps, gamma)
        return gamma, pi

    def _tuple_update_c(self, comps, cells, normal):
        ave_g = [[0j] * 3 for _ in range(3)]
        ave_p = [[0j] * 3 for _ in range(3)]
        lsum, asum = [0.0] * 3, [0.0] * 3
        bg = [[1.0 if i == j else 0.0 for j in range(3)] for i in range(3)]
        for sh, eps, name in self.dielectrics:
            lf, af = [0.0] * 3, [0.0] * 3
            for j in range(3):
                c = comps[j]
                cell = np.array(cells[j])
                lf[c] = float(self.frac_at(self.E, c, cell, name))
                af[c] = float(self.frac_at(self.D, c, cell, name))
                lsum[c] += lf[c]
                asum[c] += af[c]
            gamma, pi = self._gamma_pi_c(eps.tolist(), normal)
            for j in range(3):
                for k in range(3):
                    ave_g[j][k] = ave_g[j][k] + complex(lf[j], 0.0) * gamma[j][k]
                    ave_p[j][k] = ave_p[j][k] + complex(af[j], 0.0) * pi[j][k]
        gamma, pi = self._gamma_pi_c(bg, normal)
        for j in range(3):
            lfj, afj = 1.0 - lsum[j], 1.0 - asum[j]
            for k in range(3):
                ave_g[j][k] = ave_g[j][k] + complex(lfj, 0.0) * gamma[j][k]
                ave_p[j][k] = ave_p[j][k] + complex(afj, 0.0) * pi[j][k]
        return self._matmul3c(ave_g, self._inverse3c(ave_p))

    def _tuple_update(self, comps, cells, normal):
        """tupleUpdate, src/MxYeeFitInvEps.cpp:305-385: length-fraction weighted rows of gamma times the
        inverse of area-fraction weighted rows of pi, over the dielectrics and the background."""
        ave_g = [[0.0] * 3 for _ in range(3)]
        ave_p = [[0.0] * 3 for _ in range(3)]
        lsum, asum = [0.0] * 3, [0.0] * 3
        bg = [[1.0 if i == j else 0.0 for j in range(3)] for i in range(3)]
        for sh, eps, name in self.dielectrics:
            lf, af = [0.0] * 3, [0.0] * 3
            for j in range(3):
                c = comps[j]
                cell = np.array(cells[j])
                lf[c] = float(self.frac_at(self.E, c, cell, name))
                af[c] = float(self.frac_at(self.D, c, cell, name))
                lsum[c] += lf[c]
                asum[c] += af[c]
            gamma, pi = self._gamma_pi(eps.tolist(), normal)
            for j in range(3):
                for k in range(3):
                    ave_g[j][k] += lf[j] * gamma[j][k]
                    ave_p[j][k] += af[j] * pi[j][k]
        gamma, pi = self._gamma_pi(bg, normal)
        for j in range(3):
            lfj, afj = 1.0 - lsum[j], 1.0 - asum[j]
            for k in range(3):
                ave_g[j][k] += lfj * gamma[j][k]
                ave_p[j][k] += afj * pi[j][k]
        return geo._matmul3(ave_g, geo._inverse3(ave_p))

    def _pml_transform(self, coord, inverse):
        """Product of getInvS (inverse) / getS over the layers whose shape function is positive at `coord`
        (src/MxYeeFitInvEps.cpp:514-537, src/MxYeeFitMu.cpp:385-410); returns (T, det T, all layers diagonal)."""
        t = [[complex(1.0 if i == j else 0.0, 0.0) for j in range(3)] for i in range(3)]
        diag = True
        pt = np.array([[float(coord[0]), float(coord[1]), float(coord[2])]])
        for pml in self.pmls:
            if float(pml.shape.func(pt)[0]) > 0:
                if not pml.is_diag:
                    diag = False
                t = self._matmul3c(t, pml.inv_s(coord) if inverse else pml.s(coord))
        return t, self._det3c(t), diag

    @staticmethod
    def _det3c(m):
        """det(MxDimMatrix<T, 3>), src/MxDimMatrix.hpp:376-384."""
        return (m[0][0] * m[1][1] * m[2][2] + m[0][1] * m[1][2] * m[2][0] + m[0][2] * m[1][0] * m[2][1] -
                m[0][2] * m[1][1] * m[2][0] - m[0][1] * m[1][0] * m[2][2] - m[0][0] * m[1][2] * m[2][1])

    def _pml_sandwich(self, t, det, m):
        """(T m T) / det with the matrix / scalar quotient as m * (1 / det) (src/MxDimMatrix.hpp:152-154)."""
        r = self._matmul3c(self._matmul3c(t, m), t)
        inv = self._cdiv(complex(1.0, 0.0), det)
        return [[inv * r[i][j] for j in range(3)] for i in range(3)]

    def inv_eps(self):
        """MxYeeFitInvEps::setMatrix, src/MxYeeFitInvEps.cpp:421-560 (3-D, real eps, no PML): rows
        and columns on the E map; 1/eps_cc on rows wholly inside one (diagonal) material, otherwise
        the 9-point stencil averaged over the admissible tuples.  Entries whose column is not in
        the map are left out (Epetra never imports a value for them)."""
        E, D = self.E, self.D
        rows, cols, vals = [], [], []
        tuple_inds = [(0, 1, 5), (0, 1, 6), (0, 2, 5), (0, 2, 6), (0, 3, 7), (0, 3, 8), (0, 4, 7), (0, 4, 8)]
        names = [name for _, _, name in self.dielectrics]
        lf0 = {nm: np.empty(len(E.gids)) for nm in names}
        af0 = {nm: np.empty(len(E.gids)) for nm in names}
        for c in range(3):
            sel = E.comp == c
            for nm in names:
                lf0[nm][sel] = self.frac_at(E, c, E.cells[sel], nm)
                af0[nm][sel] = self.frac_at(D, c, E.cells[sel], nm)
        for r in range(len(E.gids)):
            comp0 = int(E.comp[r]); cell = E.cells[r]
            in_diel, eps_diag, eps_here = False, False, None
            for sh, eps, nm in self.dielectrics:                       # :462-479
                l0, a0 = lf0[nm][r], af0[nm][r]
                if l0 == 1 and a0 == 1:
                    in_diel, eps_diag, eps_here = True, self._eps_is_diag(eps), eps
                    break
                elif l0 == 0 and a0 == 0:
                    continue
                else:
                    in_diel = True
                    break
            if not in_diel:
                eps_diag, eps_here = True, np.eye(3)
            pml_diag = True
            if self.pmls:                                               # :514-537
                tr, det, pml_diag = self._pml_transform(self.comp_coord(E, comp0, cell), True)
            if eps_diag and pml_diag:                                   # :542-549
                fac = self.factor(D, comp0, cell[None, :])[0]
                ecc = eps_here[comp0, comp0]
                if self.pmls:
                    v = self._cdiv(self._cdiv(complex(fac) * tr[comp0][comp0] * tr[comp0][comp0], det), complex(ecc))
                else:
                    v = self._cdiv(fac, ecc) if self.lossy else complex(fac) / float(ecc.real)
                rows.append(r); cols.append(r); vals.append(v)
                continue
            comp1, comp2 = (comp0 + 1) % 3, (comp0 + 2) % 3            # getStencil, :35-71
            scomps = [comp0] + [comp1] * 4 + [comp2] * 4
            scells = [cell.copy() for _ in range(9)]
            scells[1][comp1] -= 1
            scells[3][comp0] += 1; scells[3][comp1] -= 1
            scells[4][comp0] += 1
            scells[5][comp2] -= 1
            scells[7][comp0] += 1; scells[7][comp2] -= 1
            scells[8][comp0] += 1
            normal = self._stencil_normal(comp0, cell, scomps, scells)
            tuples = []                                                 # getTuples, :127-175
            for t in tuple_inds:
                ok = True
                if self.pec is not None:
                    for si in t:
                        if float(self.frac_at(E, scomps[si], scells[si])) == 0:
                            ok = False
                if ok:
                    tuples.append(t)
            svals = [0.0] * 9
            nt = float(len(tuples))
            for t in tuples:                                            # :519-531
                tc, tl = [scomps[si] for si in t], [scells[si] for si in t]
                m = self._tuple_update_c(tc, tl, normal) if self.lossy else self._tuple_update(tc, tl, normal)
                if self.pmls:                                           # :567
                    m = self._pml_sandwich(tr, det, m)
                for si in t:
                    svals[si] += m[comp0][scomps[si]] / nt
            for si in range(9):                                         # :533-537
                fac = complex(self.factor(D, scomps[si], scells[si][None, :])[0])
                g = self.gid(D, scomps[si], scells[si][None, :])
                pos, ok = self.lid(D, g)
                if ok[0]:
                    rows.append(r); cols.append(int(pos[0])); vals.append(svals[si] * fac)
        n = len(E.gids)
        vals = np.array(vals, dtype=complex)
        if not self.is_complex:
            vals = vals.real
        m = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
        m.sum_duplicates(); m.sort_indices()
        return m

    def _stencil_normal(self, comp0, cell, scomps, scells):
        """getNormal, src/MxYeeFitInvEps.cpp:254-302: the (sign-aligned, summed, normalised) interface
        normals of the dielectrics that cut any member of the stencil, taken at the row's edge centre."""
        import math
        norms = []
        coord = self.comp_coord(self.E, comp0, cell)
        for sh, eps, nm in self.dielectrics:
            cut = False
            for c, cl in zip(scomps, scells):
                lf = float(self.frac_at(self.E, c, cl, nm)); af = float(self.frac_at(self.D, c, cl, nm))
                if (lf != 0 and lf != 1) or (af != 0 and af != 1):
                    cut = True
            if cut:
                g = sh.grad1((float(coord[0]), float(coord[1]), float(coord[2])))
                nn = math.sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2])
                norms.append([g[0] / nn, g[1] / nn, g[2] / nn])
        if not norms:
            return [1.0, 0.0, 0.0]
        tot = [0.0, 0.0, 0.0]
        for i, v in enumerate(norms):
            sgn = 1.0
            if i > 0 and (norms[0][0] * v[0] + norms[0][1] * v[1] + norms[0][2] * v[2]) < 0:
                sgn = -1.0
            for k in range(3):
                tot[k] += sgn * v[k]
        nn = math.sqrt(tot[0] * tot[0] + tot[1] * tot[1] + tot[2] * tot[2])
        return [tot[0] / nn, tot[1] / nn, tot[2] / nn]

    def _mu_tuple_update(self, comps, cells, normal):
        """tupleUpdate of MxYeeFitMu with mInvert = false, src/MxYeeFitMu.cpp:205-290: the inverse-permittivity averages
        with H (dual-edge length fractions) and B (face area fractions) in the roles of E and D, and the product the
        other way round: avePi * aveGamma^-1 maps the H tuple to the B tuple."""
        ave_g = [[0.0] * 3 for _ in range(3)]
        ave_p = [[0.0] * 3 for _ in range(3)]
        lsum, asum = [0.0] * 3, [0.0] * 3
        bg = [[1.0 if i == j else 0.0 for j in range(3)] for i in range(3)]
        for sh, mu, name in self.mus:
            lf, af = [0.0] * 3, [0.0] * 3
            for j in range(3):
                c = comps[j]
                cell = np.array(cells[j])
                lf[c] = float(self.frac_at(self.H, c, cell, name))
                af[c] = float(self.frac_at(self.B, c, cell, name))
                lsum[c] += lf[c]
                asum[c] += af[c]
            gamma, pi = self._gamma_pi(mu.tolist(), normal)
            for j in range(3):
                for k in range(3):
                    ave_g[j][k] += lf[j] * gamma[j][k]
                    ave_p[j][k] += af[j] * pi[j][k]
        gamma, pi = self._gamma_pi(bg, normal)
        for j in range(3):
            lfj, afj = 1.0 - lsum[j], 1.0 - asum[j]
            for k in range(3):
                ave_g[j][k] += lfj * gamma[j][k]
                ave_p[j][k] += afj * pi[j][k]
        return geo._matmul3(ave_p, geo._inverse3(ave_g))

    def _mu_tuple_update_c(self, comps, cells, normal):
        """_mu_tuple_update in complex arithmetic (the reference always computes it in MxComplex, :205-290)."""
        ave_g = [[0j] * 3 for _ in range(3)]
        ave_p = [[0j] * 3 for _ in range(3)]
        lsum, asum = [0.0] * 3, [0.0] * 3
        bg = [[1.0 if i == j else 0.0 for j in range(3)] for i in range(3)]
        for sh, mu, name in self.mus:
            lf, af = [0.0] * 3, [0.0] * 3
            for j in range(3):
                c = comps[j]
                cell = np.array(cells[j])
                lf[c] = float(self.frac_at(self.H, c, cell, name))
                af[c] = float(self.frac_at(self.B, c, cell, name))
                lsum[c] += lf[c]
                asum[c] += af[c]
            gamma, pi = self._gamma_pi_c(mu.tolist(), normal)
            for j in range(3):
                for k in range(3):
                    ave_g[j][k] = ave_g[j][k] + complex(lf[j], 0.0) * gamma[j][k]
                    ave_p[j][k] = ave_p[j][k] + complex(af[j], 0.0) * pi[j][k]
        gamma, pi = self._gamma_pi_c(bg, normal)
        for j in range(3):
            lfj, afj = 1.0 - lsum[j], 1.0 - asum[j]
            for k in range(3):
                ave_g[j][k] = ave_g[j][k] + complex(lfj, 0.0) * gamma[j][k]
                ave_p[j][k] = ave_p[j][k] + complex(afj, 0.0) * pi[j][k]
        return self._matmul3c(ave_p, self._inverse3c(ave_g))

    def mu_op(self):
        """MxYeeFitMu::setMatrix with mInvert = false, src/MxYeeFitMu.cpp:294-463 (3-D, real mu, no PML): rows and
        columns on the B map; mu_cc on rows wholly inside one (diagonal) material, otherwise the 9-point stencil of
        getStencil (:27-71) averaged over the admissible tuples.  Columns that are not in the map are left out."""
        B, H = self.B, self.H
        rows, cols, vals = [], [], []
        tuple_inds = [(0, 1, 5), (0, 1, 6), (0, 2, 5), (0, 2, 6), (0, 3, 7), (0, 3, 8), (0, 4, 7), (0, 4, 8)]
        names = [name for _, _, name in self.mus]
        lf0 = {nm: np.empty(len(B.gids)) for nm in names}
        af0 = {nm: np.empty(len(B.gids)) for nm in names}
        for c in range(3):
            sel = B.comp == c
            for nm in names:
                lf0[nm][sel] = self.frac_at(H, c, B.cells[sel], nm)
                af0[nm][sel] = self.frac_at(B, c, B.cells[sel], nm)
        for r in range(len(B.gids)):
            comp0 = int(B.comp[r]); cell = B.cells[r]
            in_mu, mu_diag, mu_here = False, False, None
            for sh, mu, nm in self.mus:                                 # :333-358
                l0, a0 = lf0[nm][r], af0[nm][r]
                if l0 == 1 and a0 == 1:
                    in_mu, mu_diag, mu_here = True, self._eps_is_diag(mu), mu
                    break
                elif l0 == 0 and a0 == 0:
                    continue
                else:
                    in_mu = True
                    break
            if not in_mu:
                mu_diag, mu_here = True, np.eye(3)
            pml_diag = True
            if self.pmls:                                               # :385-410 (mInvert = false: getS)
                tr, det, pml_diag = self._pml_transform(self.comp_coord(H, comp0, cell), False)
            if mu_diag and pml_diag:                                    # :413-424
                fac = self.factor(B, comp0, cell[None, :])[0]
                v = complex(fac) * (complex(mu_here[comp0, comp0]) if self.mu_lossy else float(mu_here[comp0, comp0].real))
                if self.pmls:
                    v = v * self._cdiv(tr[comp0][comp0] * tr[comp0][comp0], det)
                rows.append(r); cols.append(r); vals.append(v)
                continue
            comp1, comp2 = (comp0 + 1) % 3, (comp0 + 2) % 3            # getStencil, :27-71
            scomps = [comp0] + [comp1] * 4 + [comp2] * 4
            scells = [cell.copy() for _ in range(9)]
            scells[1][comp0] -= 1
            scells[2][comp0] -= 1; scells[2][comp1] += 1
            scells[4][comp1] += 1
            scells[5][comp0] -= 1
            scells[6][comp0] -= 1; scells[6][comp2] += 1
            scells[8][comp2] += 1
            normal = self._mu_stencil_normal(comp0, cell, scomps, scells)
            tuples = []                                                 # getTuples, :75-128
            for t in tuple_inds:
                ok = True
                if self.pec is not None:
                    for si in t:
                        if float(self.frac_at(B, scomps[si], scells[si])) == 0:
                            ok = False
                if ok:
                    tuples.append(t)
            svals = [0.0] * 9
            nt = float(len(tuples))
            for t in tuples:                                            # :427-440
                tc, tl = [scomps[si] for si in t], [scells[si] for si in t]
                m = self._mu_tuple_update_c(tc, tl, normal) if (self.pmls or self.mu_lossy) else self._mu_tuple_update(tc, tl, normal)
                if self.pmls:                                           # :434
                    m = self._pml_sandwich(tr, det, m)
                for si in t:
                    svals[si] += m[comp0][scomps[si]] / nt
            for si in range(9):                                         # :441-442
                fac = complex(self.factor(B, scomps[si], scells[si][None, :])[0])
                g = self.gid(B, scomps[si], scells[si][None, :])
                pos, ok = self.lid(B, g)
                if ok[0]:
                    rows.append(r); cols.append(int(pos[0])); vals.append(svals[si] * fac)
        n = len(B.gids)
        vals = np.array(vals, dtype=complex)
        if not self.is_complex:
            vals = vals.real
        m = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
        m.sum_duplicates(); m.sort_indices()
        return m

    def _mu_stencil_normal(self, comp0, cell, scomps, scells):
        """getNormal, src/MxYeeFitMu.cpp:134-186: as for the dielectrics, with the mu objects and the H / B fractions,
        taken at the row's face centre."""
        import math
        norms = []
        coord = self.comp_coord(self.B, comp0, cell)
        for sh, mu, nm in self.mus:
            cut = False
            for c, cl in zip(scomps, scells):
                lf = float(self.frac_at(self.H, c, cl, nm)); af = float(self.frac_at(self.B, c, cl, nm))
                if (lf != 0 and lf != 1) or (af != 0 and af != 1):
                    cut = True
            if cut:
                g = sh.grad1((float(coord[0]), float(coord[1]), float(coord[2])))
                nn = math.sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2])
                norms.append([g[0] / nn, g[1] / nn, g[2] / nn])
        if not norms:
            return [1.0, 0.0, 0.0]
        tot = [0.0, 0.0, 0.0]
        for i, v in enumerate(norms):
            sgn = 1.0
            if i > 0 and (norms[0][0] * v[0] + norms[0][1] * v[1] + norms[0][2] * v[2]) < 0:
                sgn = -1.0
            for k in range(3):
                tot[k] += sgn * v[k]
        nn = math.sqrt(tot[0] * tot[0] + tot[1] * tot[1] + tot[2] * tot[2])
        return [tot[0] / nn, tot[1] / nn, tot[2] / nn]

    def inv_mu_vol_ave(self):
        """cellAveInvMuOperator, src/MxYeeFitMu.cpp:532-600: volume-fraction average of 3/trace(mu) on the Psi map."""
        P = self.P
        out = np.zeros(len(P.gids)); sums = np.zeros(len(P.gids))
        for sh, mu, nm in ([] if self.mu_lossy else self.mus):
            fr = self.frac_at(P, 0, P.cells, nm)
            sums = sums + fr
            out = out + fr * (3.0 / (mu[0, 0] + mu[1, 1] + mu[2, 2]))
        out = out + (1.0 - sums) * (3.0 / 3.0)
        if self.mu_lossy:        # complex permeability: scalar loop with the explicit quotient, as inv_eps_vol_ave
            fracs = [self.frac_at(P, 0, P.cells, nm) for _, _, nm in self.mus]
            scal = [self._cdiv(complex(3.0, 0.0), complex(mu[0, 0]) + complex(mu[1, 1]) + complex(mu[2, 2])) for _, mu, _ in self.mus]
            out = np.zeros(len(P.gids), dtype=complex)
            for r in range(len(P.gids)):
                acc, sm = 0j, 0.0
                for fr, sc in zip(fracs, scal):
                    f = float(fr[r])
                    sm = sm + f
                    acc = acc + complex(f, 0.0) * sc
                out[r] = acc + complex((1.0 - sm) * (3.0 / 3.0), 0.0)
        if self.is_complex:
            out = out.astype(complex)
        # no PML 'volume' factor: the reference compiles that block out (src/MxYeeFitMu.cpp:764-778 is under `#if 0`);
        # only inv_eps_vol_ave carries det(invS) (src/MxYeeFitInvEps.cpp:779-791)
        return sp.diags(out, format="csr")

    def inv_eps_vol_ave(self):
        """cellAveInvEpsOperator, src/MxYeeFitInvEps.cpp:700-790: volume-fraction average of
        3/trace(eps) on the Psi map (diagonal)."""
        P = self.P
        if not self.lossy:
            out = np.zeros(len(P.gids)); sums = np.zeros(len(P.gids))
            for sh, eps, nm in self.dielectrics:
                fr = self.frac_at(P, 0, P.cells, nm)
                sums = sums + fr
                out = out + fr * (3.0 / (eps[0, 0] + eps[1, 1] + eps[2, 2]))
            out = out + (1.0 - sums) * (3.0 / 3.0)
            if self.is_complex:
                out = out.astype(complex)
            return sp.diags(out, format="csr")
        # complex permittivity: scalar loop with the explicit quotient (no vectorised complex arithmetic)
        fracs = [self.frac_at(P, 0, P.cells, nm) for _, _, nm in self.dielectrics]
        scal = [self._cdiv(complex(3.0, 0.0), complex(eps[0, 0]) + complex(eps[1, 1]) + complex(eps[2, 2]))
                for _, eps, _ in self.dielectrics]
        out = np.zeros(len(P.gids), dtype=complex)
        for r in range(len(P.gids)):
            acc, sums = 0j, 0.0
            for fr, sc in zip(fracs, scal):
                f = float(fr[r])
                sums = sums + f
                acc = acc + complex(f, 0.0) * sc
            out[r] = acc + complex((1.0 - sums) * (3.0 / 3.0), 0.0)
            if self.pmls:                                      # :779-791: det of the getS product
                out[r] = out[r] * self._pml_transform(self.comp_coord(P, 0, P.cells[r]), False)[1]
        return sp.diags(out, format="csr")

    # ---------------------------------------------------------------- MxMagWaveOp matrices
    def single_ops(self):
        """MxEMOps::setSingleOps, src/MxEMOps.cpp:39-168 (vacuum / PEC subset)."""
        ops = {"curlE": self.curlE(), "curlB": self.curlB(),
               "divB": self.divB(), "gradPsi": self.gradPsi()}
        if self.pec is not None:
            ops["dmA"] = self.fracs(self.B)
            ops["dmL"] = self.fracs(self.E)
            ops["dmVInv"] = self.fracs(self.P, inverse=True)
        if self.dielectrics or self.pmls:                      # src/MxEMOps.cpp:66-77
            ops["invEps"] = self.inv_eps()
            ops["invEpsVolAve"] = self.inv_eps_vol_ave()
        if self.mus or self.pmls:                              # src/MxEMOps.cpp:128-141
            ops["mu"] = self.mu_op()
            ops["invMuVolAve"] = self.inv_mu_vol_ave()
        return ops

    @staticmethod
    def multiply_ops(ops, names):
        """MxEMOps::multiplyOps -> MxUtil::Trilinos::massiveCrsMultiply (src/MxEMOps.cpp:197-225,
        src/MxUtil.cpp:318-371): the first name is applied first and the product is built as
        m_k * ( ... * (m_1 * m_0)), each step a full SpGEMM."""
        res = ops[names[0]]
        for nm in names[1:]:
            res = (ops[nm] @ res).tocsr()
            res.sum_duplicates()
            res.sort_indices()
        return res

    def wave_ops(self):
        """MxMagWaveOp::initMatrices, src/MxMagWaveOp.cpp:138-245."""
        o = self.single_ops()
        has_pec = self.pec is not None
        pec = lambda name: [name] if has_pec else []
        diel = lambda name: [name] if self.dielectrics else []
        mag = lambda name: [name] if self.mus else []
        curlcurl = self.multiply_ops(o, ["curlB"] + diel("invEps") + pec("dmL") + ["curlE"])   # :145-153
        graddiv = self.multiply_ops(o, mag("mu") + pec("dmA") + ["divB"] + pec("dmVInv") + mag("invMuVolAve") +
                                    diel("invEpsVolAve") + mag("invMuVolAve") + ["gradPsi"] + mag("mu") + pec("dmA"))  # :156-179
        scalapl = -self.multiply_ops(o, ["gradPsi"] + mag("mu") + pec("dmA") + ["divB"])    # :208-220
        if has_pec and self.mus:
            rhs = self.multiply_ops(o, ["mu", "dmA"])                                       # :224-230
        elif has_pec:
            rhs = o["dmA"]                                                                  # :233-236
        elif self.mus:
            rhs = o["mu"]                                                                   # :237-238
        else:
            rhs = sp.identity(len(self.B.gids), format="csr",
                              dtype=complex if self.is_complex else float)                 # :240-241
        veclapl = purge_zeros((curlcurl - graddiv).tocsr())                                 # :183-193
        for m in (curlcurl, graddiv, scalapl):
            m.sort_indices()
        o.update(curlCurl=curlcurl.tocsr(), gradDiv=graddiv.tocsr(), vecLapl=veclapl,
                 scaLapl=scalapl.tocsr(), mRhs=rhs.tocsr())
        return o


def purge_zeros(m, tol=1.e-12):
    """MxCrsMatrix::purgeZeros, src/MxCrsMatrix.cpp:84-116: keep |a| > 1e-12."""
    m = m.tocsr().copy()
    m.sum_duplicates()
    m.data[np.abs(m.data) <= tol] = 0.0
    m.eliminate_zeros()
    m.sort_indices()
    return m


def shifted(veclapl, rhs, shift):
    """MxMagWaveOp::setShift, src/MxMagWaveOp.cpp:248-255: vecLapl - shift * mRhs."""
    m = (veclapl - shift * rhs).tocsr()
    m.sum_duplicates()
    m.sort_indices()
    return m


def default_shift(size):
    """src/mx.py:711-712."""
    return 0.05 * (2.0 * np.pi / max(size)) ** 2


def pillbox_sim(N, **kw):
    """Config 1/2/4 input: example/pillbox.py with resolution N (unit box at -0.5)."""
    return YeeSim(N, origin=(-0.5,) * 3, size=(1.0,) * 3, pec=geo.pillbox(0.4, 0.8), **kw)


def mag_wave_modes(ops, shift, want, used=None):
    """Dense restatement of what MxSolver iterates on: T = P (L - sigma M)^-1 M with the
    divergence-cleaning projection P = I + G S^+ D M (MxMagWaveOp::Apply, src/MxMagWaveOp.cpp:826-943:
    solve with shiftedVecLapl, then scaLapl psi = divB mRhs b, b += gradPsi psi), restricted to the
    B rows MxSolver leaves in its start vector (`used`; zeroUnusedComponents, src/MxSolver.cpp:65).
    Returns k^2 = 1/theta + sigma for the `want` largest |theta| (eigValsToFreqs, :1253-1270).
    Small problems only (dense algebra)."""
    L = ops["vecLapl"].toarray(); M = ops["mRhs"].toarray()
    D = ops["divB"].toarray(); G = ops["gradPsi"].toarray(); S = ops["scaLapl"].toarray()
    n = L.shape[0]
    if used is None:
        used = (M.diagonal() > 0) & (np.abs(ops["curlE"]).sum(axis=1).A1 > 0)
    idx = np.nonzero(used)[0]
    A = (L - shift * M)[np.ix_(idx, idx)]
    T = np.zeros((n, n), dtype=np.result_type(L.dtype, M.dtype))
    T[np.ix_(idx, idx)] = np.linalg.solve(A, M[np.ix_(idx, idx)])
    P = np.eye(n) + G @ np.linalg.pinv(S, rcond=1e-12) @ D @ M
    T = (P @ T)[np.ix_(idx, idx)]
    theta = np.linalg.eigvals(T)
    theta = theta[np.argsort(-np.abs(theta))][:want]
    k2 = 1.0 / theta + shift
    return k2[np.argsort(k2.real)].real if np.abs(k2.imag).max() <= 1e-9 * np.abs(k2).max() else k2[np.argsort(k2.real)]
