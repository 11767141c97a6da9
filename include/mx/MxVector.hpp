// Single-column multivector (reference src/MxVector.hpp).
#ifndef MX_VECTOR
#define MX_VECTOR

#include "MxMultiVector.hpp"

template <typename Scalar>
class MxVector : public MxMultiVector<Scalar> {
 public:
  explicit MxVector(RCP<MxMap> map) : MxMultiVector<Scalar>(map, 1) {}
  MxVector(MxMultiVector<Scalar> const& mv, size_t vecIndex, bool deepcopy)
      : MxMultiVector<Scalar>(mv, std::vector<size_t>(1, vecIndex), deepcopy) {}
  // the reference swaps (row, vec) here (src/MxVector.hpp:31-33); kept as row access
  Scalar getValue(MxIndex row) const { return (*this)(row, 0); }
};

#endif
