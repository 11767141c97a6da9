// name -> operator registry (reference src/MxEMOps.h:20-113, src/MxEMOps.cpp:39-268).
#ifndef MX_EM_OPS
#define MX_EM_OPS

#include "MxCrsMatrix.hpp"
#include "MxEMSim.h"

template <size_t DIM, typename Scalar>
class MxEMOps {
 public:
  explicit MxEMOps(RCP<MxEMSim<DIM> > sim, bool initSingleOps = true) : pid(sim->getGrid().getComm()->myPID()), mSim(sim) {
    if (initSingleOps) setSingleOps();
  }
  // primitive operators straight from the assembler (src/MxEMOps.cpp:39-168)
  void setSingleOps() {
    const mx::YeeAssembler<Scalar>& a = mSim->assemblerFor((Scalar*)0);
    RCP<MxMap> b = mSim->getField("bfield").getMap(), e = mSim->getField("efield").getMap(),
               p = mSim->getField("psifield").getMap();
    insert("curlE", b, e, a.curlE());
    insert("curlB", e, b, a.curlB());
    if (mSim->hasPEC()) {
      insert("dmA", b, b, a.fracs(a.bfield(), false));
      insert("dmL", e, e, a.fracs(a.efield(), false));
    }
    insert("divB", p, b, a.divB());
    insert("gradPsi", b, p, a.gradPsi());
    if (mSim->hasPEC()) insert("dmVInv", p, p, a.fracs(a.psifield(), true));
    if (mSim->hasDielectric() || mSim->hasPML()) {                 // src/MxEMOps.cpp:66-77
      if (pid == 0) std::cout << "Constructing FIT inverse epsilon operator...\n";
      insert("invEps", e, e, a.invEps());
      insert("invEpsVolAve", p, p, a.invEpsVolAve());
    }
    if (mSim->hasMu() || mSim->hasPML()) {                         // src/MxEMOps.cpp:128-141
      if (pid == 0) std::cout << "Constructing FIT inverse mu operator...\n";
      insert("mu", b, b, a.muOp());
      insert("invMuVolAve", p, p, a.invMuVolAve());
    }
  }
  RCP<MxCrsMatrix<Scalar> > getOp(std::string name) const {
    auto it = mOps.find(name);
    if (it == mOps.end() && mSource && mSource->count(name)) {
      // first use of an operator of the prebuilt chain: adopt it now (on a decomposed grid this agrees its ghost
      // import, a collective - every rank asks for the same operators in the same order)
      const_cast<MxEMOps*>(this)->adopt(name, mSource->at(name));
      it = mOps.find(name);
    }
    if (it == mOps.end()) { std::cout << "MxEMOps::getOp: operator '" << name << "' not found.\n"; exit(EXIT_FAILURE); }
    return it->second;
  }
  // operators assembled elsewhere (MxEMSim::prebuiltOps) that getOp adopts on first use
  void setSource(const std::map<std::string, mx::Csr<Scalar> >* src) { mSource = src; }
  // product of the named operators, FIRST NAME APPLIED FIRST (src/MxEMOps.h:61-74):
  // result = names[n-1] * ... * names[1] * names[0]  (massiveCrsMultiply, src/MxUtil.cpp:318-371)
  RCP<MxCrsMatrix<Scalar> > multiplyOps(std::vector<std::string> const& names, std::vector<bool> const& transposes,
                                         bool fillComplete, bool store = false, std::string name = "") {
    RCP<MxCrsMatrix<Scalar> > m1 = getOp(names[0]);
    bool t1 = transposes[0];
    if (names.size() == 1) return m1;
    RCP<MxCrsMatrix<Scalar> > res;
    for (size_t i = 1; i < names.size(); ++i) {
      RCP<MxCrsMatrix<Scalar> > m2 = getOp(names[i]);
      res = rcp(new MxCrsMatrix<Scalar>(transposes[i] ? m2->getDomainMap() : m2->getRangeMap()));
      MxCrsMatrix<Scalar>::matrixMatrixMultiply(*m2, transposes[i], *m1, t1, *res, fillComplete);
      m1 = res;
      t1 = false;
    }
    if (store) mOps[name] = res;
    return res;
  }
  RCP<MxCrsMatrix<Scalar> > addOps(std::vector<std::string> const& names, std::vector<Scalar> const& scalars,
                                    bool store = false, std::string name = "") {
    RCP<MxCrsMatrix<Scalar> > acc = getOp(names[0]);
    Scalar sa = scalars[0];
    RCP<MxCrsMatrix<Scalar> > res;
    for (size_t i = 1; i < names.size(); ++i) {
      RCP<MxCrsMatrix<Scalar> > b = getOp(names[i]);
      res = rcp(new MxCrsMatrix<Scalar>(acc->getRangeMap()));
      MxCrsMatrix<Scalar>::matrixMatrixAdd(*acc, false, sa, *b, false, scalars[i], *res, true);
      acc = res;
      sa = ScalarTraits<Scalar>::one();
    }
    if (store) mOps[name] = res;
    return res;
  }
  // adopt an operator assembled elsewhere (MxEMSim::prebuiltOps); maps follow from the operator's role
  void adopt(std::string const& name, mx::Csr<Scalar> const& m) {
    RCP<MxMap> b = mSim->getField("bfield").getMap(), e = mSim->getField("efield").getMap(),
               p = mSim->getField("psifield").getMap();
    RCP<MxMap> rows = b, cols = b;
    if (name == "curlE") { cols = e; }
    else if (name == "curlB") { rows = e; }
    else if (name == "divB") { rows = p; }
    else if (name == "gradPsi") { cols = p; }
    else if (name == "dmL" || name == "invEps") { rows = e; cols = e; }
    else if (name == "dmVInv" || name == "invEpsVolAve" || name == "invMuVolAve" || name == "scaLapl") { rows = p; cols = p; }
    if (!mSim->isDistributed()) { mOps[name] = rcp(new MxCrsMatrix<Scalar>(rows, cols, &m)); return; }   // borrowed: no copy
    mOps[name] = distributed(rows, cols, fieldOf(rows, b, e), fieldOf(cols, b, e), m);
  }
  // the owned rows of a window-level operator (box + halo numbering) as a row-partitioned matrix; COLLECTIVE
  RCP<MxCrsMatrix<Scalar> > distributed(RCP<MxMap> rows, RCP<MxMap> cols, const mx::YeeField& rowF, const mx::YeeField& colF,
                                        mx::Csr<Scalar> const& m) const {
    const mx::YeeAssembler<Scalar>& a = mSim->assemblerFor((Scalar*)0);
    mx::Csr<Scalar> o = a.ownedRows(m, rowF);
    mx::UVec<int64_t> colg(o.col.size());
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < (int64_t)o.col.size(); ++q) colg[q] = colF.gids[o.col[q]];
    return rcp(new MxCrsMatrix<Scalar>(rows, cols, o.rowptr, colg, o.val));
  }
  const mx::YeeField& fieldOf(RCP<MxMap> map, RCP<MxMap> b, RCP<MxMap> e) const {
    const mx::YeeAssembler<Scalar>& a = mSim->assemblerFor((Scalar*)0);
    return map.get() == b.get() ? a.bfield() : (map.get() == e.get() ? a.efield() : a.psifield());
  }
  void saveOps() const { for (auto& e : mOps) e.second->save("Mx" + e.first); }
 private:
  void insert(const char* name, RCP<MxMap> rows, RCP<MxMap> cols, mx::Csr<Scalar> const& m) {
    if (mSim->isDistributed()) {
      std::cout << "MxEMOps: on a decomposed grid the operator chain is formed on the box + halo window (prebuilt operators)\n";
      throw 1;
    }
    mOps[name] = rcp(new MxCrsMatrix<Scalar>(rows, cols, m));
  }
  int pid;
  RCP<MxEMSim<DIM> > mSim;
  std::map<std::string, RCP<MxCrsMatrix<Scalar> > > mOps;
  const std::map<std::string, mx::Csr<Scalar> >* mSource = nullptr;
};

#endif
