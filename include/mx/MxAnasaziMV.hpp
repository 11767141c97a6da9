// Anasazi::MultiVec adapter over the device multivector (reference src/MxAnasaziMV.hpp:11-138,
// src/MxAnasaziMV.cpp:8-213).  Every virtual forwards to one HIP block kernel (K2-K8).
#ifndef MX_ANASAZI_MV
#define MX_ANASAZI_MV

#include "AnasaziCompat.hpp"
#include "MxMultiVector.hpp"

template <typename Scalar>
class MxAnasaziMV : public Anasazi::MultiVec<Scalar>, public MxMultiVector<Scalar> {
 public:
  MxAnasaziMV(RCP<MxMap> map, size_t numVecs) : MxMultiVector<Scalar>(map, numVecs) {}
  MxAnasaziMV(MxMultiVector<Scalar> const& mv) : MxMultiVector<Scalar>(mv) {}
  MxAnasaziMV(MxMultiVector<Scalar> const& mv, std::vector<size_t> const& vecInds, bool deepcopy)
      : MxMultiVector<Scalar>(mv, vecInds, deepcopy) {}

  // Clone* return raw owning pointers the caller deletes (src/MxAnasaziMV.hpp:25-29)
  virtual Anasazi::MultiVec<Scalar>* Clone(const int numVecs) const {
    return new MxAnasaziMV<Scalar>(this->getMap(), numVecs);
  }
  virtual Anasazi::MultiVec<Scalar>* CloneCopy() const { return new MxAnasaziMV<Scalar>(*this); }
  virtual Anasazi::MultiVec<Scalar>* CloneCopy(const std::vector<int>& index) const {
    return new MxAnasaziMV<Scalar>(*this, toSize(index), true);
  }
  virtual const Anasazi::MultiVec<Scalar>* CloneView(const std::vector<int>& index) const {
    return new MxAnasaziMV<Scalar>(*this, toSize(index), false);
  }
  virtual Anasazi::MultiVec<Scalar>* CloneViewNonConst(const std::vector<int>& index) {
    return new MxAnasaziMV<Scalar>(*this, toSize(index), false);
  }
  // scalar-typed global length as int (src/MxAnasaziMV.hpp:77-79)
  virtual int GetVecLength() const { return int(this->getMap()->getGlobalNumIndices()); }
  virtual int GetNumberVecs() const { return int(this->getNumVecs()); }

  // this = alpha*A*B + beta*this
  virtual void MvTimesMatAddMv(Scalar alpha, const Anasazi::MultiVec<Scalar>& A,
                               const Teuchos::SerialDenseMatrix<int, Scalar>& B, Scalar beta) {
    const MxAnasaziMV<Scalar>* Av = dynamic_cast<const MxAnasaziMV<Scalar>*>(&A);
    if (Av == NULL) { std::cout << "MxAnasaziMV::MvTimesMatAddMv cast failed\n"; exit(EXIT_FAILURE); }
    MX_ABI(mxg_mv_times_mat(mxRe(alpha), mxIm(alpha), Av->raw(), reinterpret_cast<const double*>(B.values()),
                            B.stride(), B.numRows(), B.numCols(), mxRe(beta), mxIm(beta), this->raw()),
           "MxAnasaziMV::MvTimesMatAddMv");
  }
  // this = alpha*A + beta*B (any aliasing; the reference overwrites `this` before reading B)
  virtual void MvAddMv(Scalar alpha, const Anasazi::MultiVec<Scalar>& A, Scalar beta,
                       const Anasazi::MultiVec<Scalar>& B) {
    const MxAnasaziMV<Scalar>& Av = dynamic_cast<const MxAnasaziMV<Scalar>&>(A);
    const MxAnasaziMV<Scalar>& Bv = dynamic_cast<const MxAnasaziMV<Scalar>&>(B);
    MX_ABI(mxg_mv_add(mxRe(alpha), mxIm(alpha), Av.raw(), mxRe(beta), mxIm(beta), Bv.raw(), this->raw()),
           "MxAnasaziMV::MvAddMv");
  }
  // B = alpha * A^H * this
  virtual void MvTransMv(Scalar alpha, const Anasazi::MultiVec<Scalar>& A,
                         Teuchos::SerialDenseMatrix<int, Scalar>& B) const {
    const MxAnasaziMV<Scalar>* Av = dynamic_cast<const MxAnasaziMV<Scalar>*>(&A);
    if (Av == NULL) { std::cout << "MxAnasaziMV::MvTransMv cast failed\n"; exit(EXIT_FAILURE); }
    if (B.numRows() != Av->GetNumberVecs() || B.numCols() != GetNumberVecs()) {
      std::cout << "MxAnasaziMV::MvTransMv: B is " << B.numRows() << "x" << B.numCols() << "\n";
      exit(EXIT_FAILURE);
    }
    MX_ABI(mxg_mv_trans_mv(mxRe(alpha), mxIm(alpha), Av->raw(), this->raw(), reinterpret_cast<double*>(B.values()),
                           B.stride()), "MxAnasaziMV::MvTransMv");
  }
  // B = alpha * [A A2]^H * this, one pass over the rows (mxg_mv_trans_mv2)
  virtual void MvTransMv2(Scalar alpha, const Anasazi::MultiVec<Scalar>& A, const Anasazi::MultiVec<Scalar>& A2,
                          Teuchos::SerialDenseMatrix<int, Scalar>& B) const {
    const MxAnasaziMV<Scalar>* Av = dynamic_cast<const MxAnasaziMV<Scalar>*>(&A);
    const MxAnasaziMV<Scalar>* A2v = dynamic_cast<const MxAnasaziMV<Scalar>*>(&A2);
    if (Av == NULL || A2v == NULL) { std::cout << "MxAnasaziMV::MvTransMv2 cast failed\n"; exit(EXIT_FAILURE); }
    if (B.numRows() != Av->GetNumberVecs() + A2v->GetNumberVecs() || B.numCols() != GetNumberVecs()) {
      std::cout << "MxAnasaziMV::MvTransMv2: B is " << B.numRows() << "x" << B.numCols() << "\n";
      exit(EXIT_FAILURE);
    }
    MX_ABI(mxg_mv_trans_mv2(mxRe(alpha), mxIm(alpha), Av->raw(), A2v->raw(), this->raw(), reinterpret_cast<double*>(B.values()),
                            B.stride()), "MxAnasaziMV::MvTransMv2");
  }
  virtual void MvDot(const Anasazi::MultiVec<Scalar>& A, std::vector<Scalar>& b) const {
    this->dot(dynamic_cast<const MxAnasaziMV<Scalar>&>(A), b);
  }
  virtual void MvNorm(std::vector<typename ScalarTraits<Scalar>::magnitudeType>& normvec) const {
    this->norm2(normvec);
  }
  // this[:, index[j]] = A[:, j]
  virtual void SetBlock(const Anasazi::MultiVec<Scalar>& A, const std::vector<int>& index) {
    const MxAnasaziMV<Scalar>& Av = dynamic_cast<const MxAnasaziMV<Scalar>&>(A);
    MX_ABI(mxg_mv_set_block(this->raw(), Av.raw(), index.data(), (int)std::min<size_t>(index.size(), Av.getNumVecs())),
           "MxAnasaziMV::SetBlock");
  }
  virtual void MvScale(Scalar alpha) { this->scale(alpha); }
  virtual void MvScale(const std::vector<Scalar>& alpha) { this->scale(alpha); }
  virtual void MvRandom() { this->random(); }
  virtual void MvInit(Scalar alpha) { this->set(alpha); }
  virtual void MvPrint(std::ostream& os) const { os << "MxMultiVector\n"; }

 private:
  static std::vector<size_t> toSize(const std::vector<int>& v) { return std::vector<size_t>(v.begin(), v.end()); }
};

#endif
