// Simulation container: grid + boundary conditions + PEC shape + parameter list, and the field
// maps built from them.  Light-weight counterpart of MxEMSim / MxGrid / MxGridField
// (reference src/MxEMSim.cpp:54-225, src/MxGridField.cpp:256-297): the geometry work is done by
// the Yee assembler, this class exposes what MxSolver / MxMagWaveOp ask for.
#ifndef MX_EM_SIM
#define MX_EM_SIM

#include "MxMap.hpp"
#include "MxMultiVector.hpp"
#include "mx_yee.hpp"   // maxwell_amd/csrc/host (on the include path of the host library)
#include "mxh.h"

template <size_t DIM> class MxEMSim;

template <size_t DIM>
class MxGrid {
 public:
  explicit MxGrid(RCP<MxComm> comm) : mComm(comm) {}
  RCP<MxComm> getComm() const { return mComm; }
 private:
  RCP<MxComm> mComm;
};

template <size_t DIM>
class MxGridField {
 public:
  MxGridField(std::string name, RCP<MxMap> map, std::vector<double> mask, int numComps = 3,
              std::string fieldName = "", const int* res = nullptr)
      : mName(name), mMap(map), mMask(std::move(mask)), mNumComps(numComps), mFieldName(fieldName) {
    for (int i = 0; i < 3; ++i) mRes[i] = res ? res[i] : 0;
  }
  std::string getName() const { return mName; }
  int getNumComps() const { return mNumComps; }
  // the reference's per-class field name, used in output file names (src/MxYeeFitBField.cpp:13, ...)
  std::string getFieldName() const { return mFieldName; }
  int getResolution(int i) const { return mRes[i]; }
  RCP<MxMap> getMap() const { return mMap; }
  // zero the components that are in the map but unused (zero area fraction / zero BC factor);
  // src/MxGridField.cpp "zeroUnusedComponents", called from src/MxSolver.cpp:65
  template <typename Scalar>
  static void zeroUnusedComponents(MxMultiVector<Scalar>& data, MxGridField<DIM> const& field) {
    if (field.mMask.empty()) return;
    if (!field.mMaskDev) {
      field.mMaskDev.reset(new MxMultiVector<double>(field.mMap, 1));
      field.mMaskDev->setColumn(0, field.mMask);
    }
    MX_ABI(mxg_mv_mask_rows(data.raw(), field.mMaskDev->raw()), "MxGridField::zeroUnusedComponents");
  }
 private:
  std::string mName;
  RCP<MxMap> mMap;
  std::vector<double> mMask;
  int mNumComps;
  std::string mFieldName;
  int mRes[3];
  mutable std::shared_ptr<MxMultiVector<double> > mMaskDev;
};

template <size_t DIM>
class MxEMSim {
 public:
  // A problem with Bloch phase shifts (src/MxGridField.cpp:34-38,157-188) or lossy dielectrics is assembled in complex arithmetic;
  // everything else in real arithmetic.
  MxEMSim(mx::YeeConfig const& cfg, RCP<MxComm> comm, Teuchos::ParameterList pList = Teuchos::ParameterList(),
          std::shared_ptr<mx::YeeAssembler<double> > assembler = nullptr,
          std::shared_ptr<mx::YeeAssembler<MxComplex> > complexAssembler = nullptr)
      : mCfg(cfg), mGrid(comm), mPList(pList) {
    static_assert(DIM == 3, "only the 3-D path is built");
    mComplex = cfg.phaseShifts[0] != 0.0 || cfg.phaseShifts[1] != 0.0 || cfg.phaseShifts[2] != 0.0;
    for (auto const& dl : cfg.dielectrics) if (dl.lossy()) mComplex = true;   // loss tangent, src/MxProblem.cpp:507-508
    if (!cfg.pmls.empty()) mComplex = true;                                   // src/MxProblem.cpp:596-597
    for (auto const& m : cfg.mus) if (m.lossy()) mComplex = true;              // src/MxProblem.cpp:536-537
    if (mComplex) {
      mAsmC = complexAssembler ? complexAssembler : std::make_shared<mx::YeeAssembler<MxComplex> >(cfg);
      addFields(*mAsmC, comm);
    } else {
      mAsm = assembler ? assembler : std::make_shared<mx::YeeAssembler<double> >(cfg);
      addFields(*mAsm, comm);
    }
    mPList.set("has curl null", true);
  }
  bool isComplex() const { return mComplex; }
  // true when this rank holds one MxGridDomain box of a decomposed grid
  bool isDistributed() const { return mWindowed; }
  const mx::YeeConfig& getConfig() const { return mCfg; }
  // operators already assembled for this configuration (the mxh_sim of the C ABI computes the whole chain for
  // export); MxMagWaveOp::initMatrices adopts them instead of repeating the products.  Real problems only.
  typedef std::map<std::string, mx::Csr<double> > OpCache;
  typedef std::map<std::string, mx::Csr<MxComplex> > OpCacheC;
  void setPrebuiltOps(const OpCache* ops) { mPrebuilt = ops; }
  void setPrebuiltOps(const OpCacheC* ops) { mPrebuiltC = ops; }
  const OpCache* prebuiltOps(double*) const { return mPrebuilt; }
  const OpCacheC* prebuiltOps(MxComplex*) const { return mPrebuiltC; }
  // complex problem on a decomposed grid: the window assembler is shared with whoever formed the operator chain
  void setComplexAssembler(std::shared_ptr<mx::YeeAssembler<MxComplex> > a) { mAsmC = a; }
  bool hasPEC() const { return (bool)mCfg.pec; }
  bool hasDielectric() const { return !mCfg.dielectrics.empty(); }
  bool hasMu() const { return !mCfg.mus.empty(); }
  bool hasPML() const { return !mCfg.pmls.empty(); }
  const MxGrid<DIM>& getGrid() const { return mGrid; }
  Teuchos::ParameterList getParameters() const { return mPList; }
  void setParameters(Teuchos::ParameterList p) { mPList = p; mPList.set("has curl null", true); }
  const MxGridField<DIM>& getField(std::string fieldName) const {
    auto it = mFields.find(fieldName);
    if (it == mFields.end()) { std::cout << "MxEMSim<" << DIM << ">::getField: field '" << fieldName << "' not found."; throw 1; }
    return *it->second;
  }
  const mx::YeeAssembler<double>& getAssembler() const {
    if (!mAsm) { std::cout << "MxEMSim::getAssembler: the problem is complex (phase shifts); use the complex solver\n"; throw 1; }
    return *mAsm;
  }
  // assembler in the solver's scalar type: a complex solve of a real problem assembles a complex copy on demand
  const mx::YeeAssembler<double>& assemblerFor(double*) const { return getAssembler(); }
  const mx::YeeAssembler<MxComplex>& assemblerFor(MxComplex*) const {
    if (!mAsmC) mAsmC = std::make_shared<mx::YeeAssembler<MxComplex> >(mCfg);
    return *mAsmC;
  }
 private:
  template <typename A>
  void addFields(const A& a, RCP<MxComm> comm) {
    mWindowed = a.isWindowed();
    if (mWindowed) {
      // one rank per MxGridDomain box (src/MxGrid.cpp:171-193): the box of the configuration must be this rank's
      int lo[3], hi[3];
      if (mxh_partition(mCfg.n, comm->numProc(), comm->myPID(), mProcGrid, lo, hi) != 0) throw 1;
      for (int i = 0; i < 3; ++i) {
        const int chi = mCfg.boxHi[i] < 0 ? mCfg.n[i] + 1 : mCfg.boxHi[i];
        if (lo[i] != mCfg.boxLo[i] || hi[i] != chi) {
          std::cout << "MxEMSim: the cell box of the configuration is not the box of rank " << comm->myPID() << " of "
                    << comm->numProc() << " (MxGrid::getGridDomain)\n";
          throw 1;
        }
      }
    } else if (comm->numProc() > 1) {
      std::cout << "MxEMSim: " << comm->numProc() << " ranks need a configuration restricted to this rank's cell box\n";
      throw 1;
    }
    addField("bfield", a.bfield(), a.usedMask(), comm);
    addField("efield", a.efield(), std::vector<double>(), comm);
    addField("psifield", a.psifield(), std::vector<double>(), comm);
    if (hasDielectric() || hasPML()) mFields["dfield"] = mFields["efield"];   // dfield->setMap(efield), src/MxEMSim.cpp:196-197
  }
  void addField(const char* name, const mx::YeeField& f, std::vector<double> mask, RCP<MxComm> comm) {
    std::vector<MxIndex> g;
    RCP<MxMap> map;
    if (!mWindowed) {
      g.assign(f.gids.begin(), f.gids.end());
      map = rcp(new MxMap(std::move(g), comm));
    } else {
      // owned entries only (ascending global id); ghosts are a matter of each operator (MxCrsMatrix)
      std::vector<double> m2;
      for (size_t r = 0; r < f.gids.size(); ++r)
        if (f.owned[r]) { g.push_back((MxIndex)f.gids[r]); if (!mask.empty()) m2.push_back(mask[r]); }
      mask.swap(m2);
      double total = (double)g.size();
      MX_ABI(mxg_comm_allreduce_host(comm->getCtx(), &total, 1), "MxEMSim (global map size)");
      map = rcp(new MxMap((size_t)(total + 0.5), std::move(g), comm));
      const int ncomp = f.ncomp;
      const int n[3] = {mCfg.n[0], mCfg.n[1], mCfg.n[2]}, pg[3] = {mProcGrid[0], mProcGrid[1], mProcGrid[2]};
      map->setOwnerFunction([ncomp, n, pg](MxIndex gid) {
        const int64_t cell = (int64_t)gid / ncomp, d1 = n[1] + 1, d2 = n[2] + 1;
        const int c[3] = {(int)(cell / (d2 * d1)), (int)((cell / d2) % d1), (int)(cell % d2)};
        return mxh_owner_of_cell(n, pg, c);
      });
    }
    const std::string nm(name);
    const char* cls = nm == "bfield" ? "YeeFitBField" : (nm == "efield" ? "YeeFitEField" : "YeeFitPsiField");
    mFields[name] = std::make_shared<MxGridField<DIM> >(name, map, std::move(mask), f.ncomp, cls, mCfg.n);
  }
  mx::YeeConfig mCfg;
  MxGrid<DIM> mGrid;
  Teuchos::ParameterList mPList;
  bool mComplex = false;
  bool mWindowed = false;
  int mProcGrid[3] = {1, 1, 1};
  const OpCache* mPrebuilt = nullptr;
  const OpCacheC* mPrebuiltC = nullptr;
  std::shared_ptr<mx::YeeAssembler<double> > mAsm;
  mutable std::shared_ptr<mx::YeeAssembler<MxComplex> > mAsmC;
  std::map<std::string, std::shared_ptr<MxGridField<DIM> > > mFields;
};

#endif
