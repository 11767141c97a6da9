// Yee / Dey-Mittra operator assembly on the host (setup; feeds the GPU hot path).
//
// Restates, for 3-D, the semantics of
//   lattice / indices   src/MxGrid.h:96-128, src/MxGrid.cpp:100-193, src/MxGridDomain.cpp:21-89
//   field layouts       src/MxYeeMagFieldBase.cpp:85-116, src/MxYeeElecFieldBase.cpp:64-82,
//                       src/MxYeePsiField.cpp:50-70
//   BC translation      src/MxYeeMagFieldBase.cpp:119-168, src/MxYeeElecFieldBase.cpp:89-138,
//                       src/MxYeePsiField.cpp:73-96
//   map membership      src/MxGridField.cpp:41-77,256-297, src/MxYeeFitBField.cpp:63-80,
//                       src/MxYeeFitEField.cpp:53-88, src/MxYeePsiField.cpp:100-114
//   BC mirroring/phase  src/MxGridField.cpp:80-190, src/MxYeeFitBField.cpp:83-90,
//                       src/MxYeeFitEField.cpp:91-98, src/MxYeePsiField.cpp:117-124
//   fractions           src/MxGridField.hpp:177-183, src/MxGridField.cpp:193-226
//   operators           src/MxYeeDeyMittraCurlE.cpp:117-179, CurlB.cpp:113-169, DivB.cpp:157-200,
//                       GradPsi.cpp:23-84, Fracs.cpp:30-131
//   products            src/MxMagWaveOp.cpp:138-273, src/MxEMOps.cpp:197-268, src/MxUtil.cpp:318-371
// as one assembler object instead of the reference's class-per-operator hierarchy: the hot
// path only consumes the finished CSR arrays.
#ifndef MX_YEE_HPP
#define MX_YEE_HPP

#include <complex>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "mx_geometry.hpp"
#include "mx/mx_sparse.hpp"

namespace mx {

enum BcKind { BC_PERIODIC = 0, BC_PEC = 1, BC_PMC = 2 };          // user-level (MxTypes.h)
enum CompBc { CBC_PERIODIC = 0, CBC_ZERO = 1, CBC_CONSTANT = 2 };  // per component and axis

// <Dielectric> block (src/MxProblem.cpp:487-527, src/MxDielectric.cpp): symmetric tensor; a loss tangent
// t makes the diagonal complex, eps_ii (1 + i t) (src/MxProblem.cpp:505-513) and the problem complex
struct YeeDielectric {
  std::string name;
  std::shared_ptr<const Shape> shape;
  double eps[3][3];
  double epsIm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  bool lossy() const {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) if (epsIm[i][j] != 0.0) return true;
    return false;
  }
};

// <Mu> block (src/MxProblem.cpp:515-556, src/MxMu.cpp): symmetric real permeability tensor of a shape
struct YeeMu {
  std::string name;
  std::shared_ptr<const Shape> shape;
  double mu[3][3];
  double muIm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};   // loss tangent t: mu_ii (1 + i t), src/MxProblem.cpp:536-547
  bool lossy() const {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) if (muIm[i][j] != 0.0) return true;
    return false;
  }
};

// <PML> block (src/MxProblem.cpp:559-598, src/MxPML.h): stretch s = 1 + i alphaMax ((x - begin) . dir / width)^exponent
// along dir = (end - begin) / width inside `shape`; makes the problem complex
struct YeePml {
  std::string name;
  std::shared_ptr<const Shape> shape;
  double alphaMax = 1.0, exponent = 2.0;
  double begin[3] = {0, 0, 0}, end[3] = {0, 0, 0};
};

struct YeeConfig {
  int n[3] = {16, 16, 16};
  double origin[3] = {0, 0, 0};
  double size[3] = {1, 1, 1};
  BcKind lower[3] = {BC_PERIODIC, BC_PERIODIC, BC_PERIODIC};
  BcKind upper[3] = {BC_PERIODIC, BC_PERIODIC, BC_PERIODIC};
  double phaseShifts[3] = {0, 0, 0};
  std::shared_ptr<const Shape> pec;   // null: no PEC region
  // true: keep the reference's behaviour of testing useCompInMap on the un-interiorised cell in
  // MxYeeFitEField/MxYeePsiField::getCompFactor (drops periodic wrap entries of curlE at the
  // upper boundary); false: test the interiorised cell
  bool strict = true;
  double dmFrac = 0.0;   // <PEC> 'D-M frac' (src/MxProblem.cpp:450): B faces below it get factor 0
  std::vector<YeeDielectric> dielectrics;   // in deck order; background is vacuum (src/MxEMSim.cpp:32-33)
  std::vector<YeeMu> mus;                   // magnetic materials in deck order; background mu = 1
  std::vector<YeePml> pmls;                 // perfectly matched layers in deck order
  // Distributed assembly (one GPU per MxGridDomain box, src/MxGrid.cpp:171-193): the cells this
  // rank owns, [boxLo, boxHi) per axis; boxHi[i] < 0 means the whole axis, [0, n+1).  The
  // assembler works on the box grown by `halo` cells wherever the axis is split, so the owned rows
  // of every product (curlCurl, gradDiv, vecLapl, scaLapl) are exact without any communication.
  int boxLo[3] = {0, 0, 0};
  int boxHi[3] = {-1, -1, -1};
  int halo = 2;
};

struct YeeField {
  std::string name;
  int ncomp = 0;
  bool xi0[3][3];                 // cell-relative coordinate == 0 along axis i for component c
  int lbc[3][3], ubc[3][3];       // CompBc per component and axis
  std::vector<UVec<double>> frac;   // per component over the extended lattice; empty: no region
  std::vector<std::vector<UVec<double>>> dielFrac;   // [dielectric][component]: shape reps that are not a region
  std::vector<std::vector<UVec<double>>> muFrac;     // [mu object][component]
  UVec<int64_t> gids;             // global ids of the (window-)map entries; ascending on the owned box
  UVec<int32_t> wcell;            // window cell index of every map entry
  UVec<uint8_t> owned;            // 1 if the entry's cell lies in the owned box
  UVec<int32_t> lidOfWin;         // dense lookup over ncomp * window cells, -1 if absent
};

template <typename S>
class YeeAssembler {
 public:
  explicit YeeAssembler(const YeeConfig& cfg);

  const YeeField& bfield() const { return B_; }
  const YeeField& efield() const { return E_; }
  const YeeField& psifield() const { return P_; }
  const YeeField& dfield() const { return D_; }
  bool hasDielectric() const { return !cfg_.dielectrics.empty(); }
  bool hasMu() const { return !cfg_.mus.empty(); }
  bool hasPml() const { return !cfg_.pmls.empty(); }
  bool hasPec() const { return (bool)cfg_.pec; }
  const YeeConfig& config() const { return cfg_; }
  double cellSize(int i) const { return d_[i]; }

  // primitive operators (MxEMOps::setSingleOps)
  Csr<S> curlE() const;
  Csr<S> curlB() const;
  Csr<S> divB() const;
  Csr<S> gradPsi() const;
  Csr<S> fracs(const YeeField& F, bool inverse) const;
  // FIT inverse permittivity on the E map and its cell average on the Psi map (src/MxYeeFitInvEps.cpp)
  Csr<S> invEps() const;
  Csr<S> invEpsVolAve() const;
  // FIT permeability on the B map (H -> B) and the cell average of 1/mu on the Psi map (src/MxYeeFitMu.cpp)
  Csr<S> muOp() const;
  Csr<S> invMuVolAve() const;

  // name -> matrix: curlE curlB divB gradPsi [dmA dmL dmVInv] curlCurl gradDiv vecLapl scaLapl mRhs
  std::map<std::string, Csr<S>> waveOps() const;
  std::map<std::string, Csr<S>> waveOps(std::map<std::string, Csr<S>> singles) const;   // chain from given primitives
  // the single Yee operators only (the products are then formed elsewhere: on the device, MXH_DEVICE_CHAIN)
  std::map<std::string, Csr<S>> singleOps() const;
  // massiveCrsMultiply order: first name applied first, m_k * (... * (m_1 * m_0))
  static Csr<S> multiplyOps(const std::map<std::string, Csr<S>>& ops,
                            const std::vector<std::string>& names);

  // rows of the B map whose own component factor is zero (empty rows of curlE / gradPsi)
  std::vector<double> usedMask() const;
  int64_t cellToGlobal(const int c[3]) const;
  void globalToCell(int64_t idx, int c[3]) const;

  // distributed helpers: true if the assembler works on a sub-box
  bool isWindowed() const { return windowed_; }
  // rows of the owned box only (ascending global id); columns stay window-local
  Csr<S> ownedRows(const Csr<S>& m, const YeeField& rowF) const;
  // global ids of the owned entries of a field (ascending) and of every window entry
  UVec<int64_t> ownedGids(const YeeField& F) const;

 private:
  YeeConfig cfg_;
  int n_[3];
  double d_[3];
  std::complex<double> phase_[3];
  bool complexProblem_;
  YeeField B_, E_, P_, D_;   // D: E's positions and map, represented by the dual faces (src/MxYeeFitDField.cpp:42-52)
  YeeField H_;               // H: B's positions and map, represented by the dual edges (src/MxYeeFitHField.cpp:37-47)
  UVec<uint8_t> bcellUsed_;  // extended lattice

  bool windowed_;
  bool periodicAxis_[3];
  bool splitAxis_[3];
  // canonical image of a window coordinate: on a split periodic axis, window cells beyond the seam
  // stand for the cells of the next period (cell n+1 is cell 1, node n is node 0), so everything
  // stored for them is evaluated at the wrapped coordinate; a whole axis keeps the reference's
  // lattice 0..n with its special upper node layer
  void canon(const int u[3], int c[3]) const {
    for (int i = 0; i < 3; ++i)
      c[i] = (periodicAxis_[i] && splitAxis_[i]) ? ((u[i] % n_[i]) + n_[i]) % n_[i] : u[i];
  }
  int wlo_[3], whi_[3];     // window [wlo, whi) in unwrapped cell coordinates
  // window-unwrapped coordinate of cell c (wrapped or not) inside [wlo - g, whi + g); false if outside
  bool toWin(const int c[3], int g, int u[3]) const {
    for (int i = 0; i < 3; ++i) {
      const int lo = wlo_[i] - g, hi = whi_[i] + g;
      int v = c[i];
      if (v < lo || v >= hi) {
        if (!periodicAxis_[i]) return false;
        if (v - n_[i] >= lo && v - n_[i] < hi) v -= n_[i];
        else if (v + n_[i] >= lo && v + n_[i] < hi) v += n_[i];
        else return false;
      }
      u[i] = v;
    }
    return true;
  }
  // index into the extended lattice (window + one guard cell), -1 if outside
  int64_t extIndex(const int c[3]) const {
    int u[3];
    if (!toWin(c, 1, u)) return -1;
    return ((int64_t)(u[0] - wlo_[0] + 1) * (whi_[1] - wlo_[1] + 2) + (u[1] - wlo_[1] + 1)) * (whi_[2] - wlo_[2] + 2) +
           (u[2] - wlo_[2] + 1);
  }
  int64_t extSize() const {
    return (int64_t)(whi_[0] - wlo_[0] + 2) * (whi_[1] - wlo_[1] + 2) * (whi_[2] - wlo_[2] + 2);
  }
  // index into the window lattice, -1 if outside
  int64_t winIndex(const int c[3]) const {
    int u[3];
    if (!toWin(c, 0, u)) return -1;
    return ((int64_t)(u[0] - wlo_[0]) * (whi_[1] - wlo_[1]) + (u[1] - wlo_[1])) * (whi_[2] - wlo_[2]) + (u[2] - wlo_[2]);
  }
  void winCell(int64_t idx, int u[3]) const {
    const int n1 = whi_[1] - wlo_[1], n2 = whi_[2] - wlo_[2];
    u[2] = (int)(idx % n2) + wlo_[2];
    u[1] = (int)((idx / n2) % n1) + wlo_[1];
    u[0] = (int)(idx / ((int64_t)n1 * n2)) + wlo_[0];
  }
  int32_t lidOf(const YeeField& F, int comp, const int c[3]) const {   // c: any image of the cell
    const int64_t w = winIndex(c);
    return w < 0 ? -1 : F.lidOfWin[w * F.ncomp + comp];
  }
  void interior(const YeeField& F, int comp, const int c[3], int out[3]) const;
  double fracAt(const YeeField& F, int comp, const int c[3]) const;
  bool baseUse(const YeeField& F, int comp, const int c[3]) const;
  bool bUse(const int c[3]) const;
  bool eUse(int comp, const int c[3]) const;
  bool psiUse(const int c[3]) const;
  bool use(const YeeField& F, int comp, const int c[3]) const;
  std::complex<double> baseFactor(const YeeField& F, int comp, const int c[3]) const;
  std::complex<double> factor(const YeeField& F, int comp, const int c[3]) const;
  int64_t gid(const YeeField& F, int comp, const int c[3]) const;
  void addShapeRep(const YeeField& F, const Shape& shape, std::vector<UVec<double>>& out) const;
  double dielFracAt(const YeeField& F, size_t diel, int comp, const int c[3]) const;
  double muFracAt(const YeeField& F, size_t obj, int comp, const int c[3]) const;
  void muStencilNormal(int comp0, const int cell[3], const int scomps[9], const int scells[9][3], double n[3]) const;
  void muTupleUpdate(const int comps[3], const int cells[3][3], const double n[3], double out[3][3]) const;
  void stencilNormal(int comp0, const int cell[3], const int scomps[9], const int scells[9][3], double n[3]) const;
  void tupleUpdate(const int comps[3], const int cells[3][3], const double n[3], double out[3][3]) const;
  void tupleUpdateC(const int comps[3], const int cells[3][3], const double n[3], std::complex<double> out[3][3]) const;
  void muTupleUpdateC(const int comps[3], const int cells[3][3], const double n[3], std::complex<double> out[3][3]) const;
  // product of getInvS (inverse) / getS of the layers whose shape function is positive at p; returns "all diagonal"
  bool pmlTransform(const double p[3], bool inverse, std::complex<double> t[3][3], std::complex<double>& det) const;
  void compCoord(const YeeField& F, int comp, const int cell[3], double p[3]) const;
  struct PmlGeom { double width, dir[3]; std::complex<double> outer[3][3], compl_[3][3]; };
  std::vector<PmlGeom> pmlGeom_;
  bool lossy_ = false;   // complex material arithmetic: a loss tangent or a PML
  bool muLossy_ = false; // complex permeability
  void setMap(YeeField& F);
  struct Entry { int comp; int off[3]; double val; };
  Csr<S> assemble(const YeeField& rowF, const YeeField& colF, bool skipDeadRows,
                  const std::vector<std::vector<Entry>>& stencilPerRowComp) const;
  static S toScalar(const std::complex<double>& z);
};

BcKind parseBc(const std::string& s);

}  // namespace mx
#endif
