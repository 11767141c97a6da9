// Device-resident multivector with the reference's interface (src/MxMultiVector.hpp:14-109).
// Bodies call the C ABI of mxg.h instead of Epetra_MultiVector.
#ifndef MX_MULTI_VECTOR
#define MX_MULTI_VECTOR

#include "MxMap.hpp"

template <typename Scalar> class MxCrsMatrix;
template <typename Scalar> class MxVector;

// shared ownership of one mxg_mv handle; views keep their parent's storage alive
struct MxMvHandle {
  mxg_mv* h = nullptr;
  std::shared_ptr<MxMvHandle> parent;
  explicit MxMvHandle(mxg_mv* hh, std::shared_ptr<MxMvHandle> p = nullptr) : h(hh), parent(p) {}
  ~MxMvHandle() { if (h) mxg_mv_destroy(h); }
 private:
  MxMvHandle(const MxMvHandle&);
};

template <typename Scalar>
class MxMultiVector {
 public:
  MxMultiVector(RCP<MxMap> map, size_t numVecs) : mNumVecs(numVecs), mMap(map) {
    mxg_mv* h = nullptr;
    MX_ABI(mxg_mv_create(map->getComm()->getCtx(), (int64_t)map->getNodeNumIndices(), (int)numVecs,
                         ScalarTraits<Scalar>::isComplex, &h), "MxMultiVector");
    mVec = std::make_shared<MxMvHandle>(h);
  }

  // copy ctor: deep copy (Epetra_MultiVector copy, src/MxMultiVector.cpp:48-58)
  MxMultiVector(MxMultiVector<Scalar> const& mv) : mNumVecs(mv.mNumVecs), mMap(mv.mMap) {
    mxg_mv* h = nullptr;
    MX_ABI(mxg_mv_clone(mv.raw(), NULL, 0, &h), "MxMultiVector(copy)");
    mVec = std::make_shared<MxMvHandle>(h);
  }

  // columns vecInds of mv: deep copy or aliasing view (src/MxMultiVector.cpp:29-43)
  MxMultiVector(MxMultiVector<Scalar> const& mv, std::vector<size_t> const& vecInds, bool deepcopy)
      : mNumVecs(vecInds.size()), mMap(mv.mMap) {
    std::vector<int> cols(vecInds.begin(), vecInds.end());
    mxg_mv* h = nullptr;
    if (deepcopy) {
      MX_ABI(mxg_mv_clone(mv.raw(), cols.data(), (int)cols.size(), &h), "MxMultiVector(copy cols)");
      mVec = std::make_shared<MxMvHandle>(h);
    } else {
      MX_ABI(mxg_mv_view(mv.raw(), cols.data(), (int)cols.size(), &h), "MxMultiVector(view)");
      mVec = std::make_shared<MxMvHandle>(h, mv.mVec);
    }
  }

  virtual ~MxMultiVector() {}

  size_t getNumVecs() const { return mNumVecs; }

  // global / local element writes (host round trip: setup and I/O only)
  void replaceGlobalValue(MxIndex row, size_t vec, Scalar value) {
    const MxIndex lid = mMap->getLocalIndex(row);
    if (lid != MxIndex(-1)) replaceLocalValue(lid, vec, value);
  }
  void replaceLocalValue(MxIndex row, size_t vec, Scalar value) {
    MX_ABI(mxg_mv_put(raw(), (int)vec, (int64_t)row, 1, reinterpret_cast<const double*>(&value)),
           "MxMultiVector::replaceLocalValue");
  }

  void conj() { MX_ABI(mxg_mv_conj(raw()), "MxMultiVector::conj"); }
  void scale(Scalar val) { MX_ABI(mxg_mv_scale(raw(), mxRe(val), mxIm(val)), "MxMultiVector::scale"); }
  void scale(std::vector<Scalar> const& vals) {
    MX_ABI(mxg_mv_scale_cols(raw(), reinterpret_cast<const double*>(vals.data())), "MxMultiVector::scale(cols)");
  }
  virtual void random() { MX_ABI(mxg_mv_random(raw(), nextSeed()), "MxMultiVector::random"); }
  void norm2(std::vector<double>& norms) const {
    norms.resize(mNumVecs);
    MX_ABI(mxg_mv_norm2(raw(), norms.data()), "MxMultiVector::norm2");
  }
  // v_j /= ||v_j||  (the reference divides by the inverse norm, src/MxMultiVector.cpp:163-172: a
  // bug not reproduced; normalize() is not called on the hot path)
  void normalize() {
    std::vector<double> n;
    norm2(n);
    std::vector<Scalar> inv(n.size());
    for (size_t i = 0; i < n.size(); ++i) inv[i] = Scalar(1.0 / n[i]);
    scale(inv);
  }
  void set(Scalar val) { MX_ABI(mxg_mv_fill(raw(), mxRe(val), mxIm(val)), "MxMultiVector::set"); }

  // res_j = mv_j^H . this_j (documented semantics, src/MxMultiVector.hpp:52-55; the reference's
  // complex specialisation always returns a zero imaginary part, src/MxMultiVector.cpp:196-199)
  void dot(MxMultiVector<Scalar> const& mv, std::vector<Scalar>& res) const {
    res.resize(mNumVecs);
    MX_ABI(mxg_mv_dot(mv.raw(), raw(), reinterpret_cast<double*>(res.data())), "MxMultiVector::dot");
  }
  // this = scalarA * mvA + scalarThis * this
  void update(Scalar scalarA, MxMultiVector<Scalar> const& mvA, Scalar scalarThis) {
    MX_ABI(mxg_mv_axpby(mxRe(scalarA), mxIm(scalarA), mvA.raw(), mxRe(scalarThis), mxIm(scalarThis), raw()),
           "MxMultiVector::update");
  }

  void setVector(size_t vecIndex, RCP<MxVector<Scalar> const> vec, bool copy);
  RCP<MxVector<Scalar> const> getVector(size_t vecIndex, bool copy) const;
  RCP<MxVector<Scalar> > getVectorNonConst(size_t vecIndex, bool copy);

  Scalar operator()(MxIndex localIndex, size_t vec) const {
    Scalar v;
    MX_ABI(mxg_mv_fetch(raw(), (int)vec, (int64_t)localIndex, 1, reinterpret_cast<double*>(&v)),
           "MxMultiVector::operator()");
    return v;
  }

  MxMultiVector<Scalar>& operator=(MxMultiVector<Scalar> const& mv) {
    if (this != &mv) MX_ABI(mxg_mv_copy(raw(), mv.raw()), "MxMultiVector::operator=");
    return *this;
  }

  RCP<MxMap> getMap() const { return mMap; }
  MxIndex getLocalLength() const { return mMap->getNodeNumIndices(); }

  // bulk host <-> device transfer of one column (MxIO / tests)
  void getColumn(size_t vec, std::vector<Scalar>& out) const {
    out.resize(getLocalLength());
    MX_ABI(mxg_mv_fetch(raw(), (int)vec, 0, (int64_t)out.size(), reinterpret_cast<double*>(out.data())),
           "MxMultiVector::getColumn");
  }
  void setColumn(size_t vec, std::vector<Scalar> const& in) {
    MX_ABI(mxg_mv_put(raw(), (int)vec, 0, (int64_t)in.size(), reinterpret_cast<const double*>(in.data())),
           "MxMultiVector::setColumn");
  }

  // advanced usage (the analogue of getRawMV())
  mxg_mv* raw() const { return mVec->h; }

  friend class MxCrsMatrix<Scalar>;

 protected:
  static uint64_t nextSeed() { static uint64_t s = 12345; return s++; }
  size_t mNumVecs;
  RCP<MxMap> mMap;
  std::shared_ptr<MxMvHandle> mVec;
};

#include "MxVector.hpp"

template <typename Scalar>
void MxMultiVector<Scalar>::setVector(size_t vecIndex, RCP<MxVector<Scalar> const> vec, bool) {
  std::vector<size_t> idx(1, vecIndex);
  MxMultiVector<Scalar> col(*this, idx, false);
  col = *vec;
}
template <typename Scalar>
RCP<MxVector<Scalar> const> MxMultiVector<Scalar>::getVector(size_t vecIndex, bool copy) const {
  return RCP<MxVector<Scalar> const>(new MxVector<Scalar>(*this, vecIndex, copy));
}
template <typename Scalar>
RCP<MxVector<Scalar> > MxMultiVector<Scalar>::getVectorNonConst(size_t vecIndex, bool copy) {
  return rcp(new MxVector<Scalar>(*this, vecIndex, copy));
}

#endif
