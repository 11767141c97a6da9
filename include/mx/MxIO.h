// MxIO<DIM, Scalar>::save (reference src/MxIO.h, src/MxIO.cpp:167-222): one file per vector,
// `mx<FieldName>_<name>_vecNN.h5`, dataset "real" (and "imag" for complex scalars) of shape
// (Nx+1, Ny+1, Nz+1, ncomp), zero where the component is not in the field's map.  The global id of a
// component is comp + ncomp * cellIndex with the cell index in C order over the node lattice
// (src/MxGridField.hpp:142-145, src/MxGrid.h:98-113), so the file array is the vector scattered by global id.
// Written through mxh_field_write: HDF5 when a library can be loaded at run time, NumPy `.npy` otherwise.
#ifndef MX_IO_H
#define MX_IO_H

#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "MxEMSim.h"
#include "MxMultiVector.hpp"
#include "mxh.h"

#define MX_ABI_H(call, where) do { if ((call) != 0) { std::cout << where << ": " << mxh_last_error() << "\n"; throw 1; } } while (0)

template <size_t DIM, typename Scalar>
class MxIO {
 public:
  explicit MxIO(RCP<MxComm> comm) : mComm(comm) {}

  // returns the paths written
  std::vector<std::string> save(MxMultiVector<Scalar> const& fieldData, MxGridField<DIM> const& field, std::string name) {
    std::vector<std::string> out;
    if (fieldData.getMap()->getNodeNumIndices() != field.getMap()->getNodeNumIndices()) {
      std::cout << "MxIO<DIM, Scalar>::save(fieldData, gridField, name): dist map of fieldData does not match that of gridField. Field save aborted.\n";
      return out;
    }
    const int nc = field.getNumComps();
    int64_t dims[4] = {field.getResolution(0) + 1, field.getResolution(1) + 1, field.getResolution(2) + 1, nc};
    const size_t total = (size_t)(dims[0] * dims[1] * dims[2] * dims[3]);
    const MxMap& map = *field.getMap();
    const size_t nloc = map.getNodeNumIndices();
    // decomposed grid: rank 0 collects every rank's (global id, value) pairs and writes the whole lattice, like the
    // single file per vector the reference's parallel HDF5 write produces (src/MxIO.cpp:167-222)
    const int np = mComm->numProc(), me = mComm->myPID();
    std::vector<int64_t> gids(nloc), counts(np, 0), allGids;
    for (size_t l = 0; l < nloc; ++l) gids[l] = (int64_t)map.getGlobalIndex(l);
    if (np > 1) {
      MX_ABI(mxg_comm_allgather_i64(mComm->getCtx(), (int64_t)(nloc * sizeof(int64_t)), counts.data()), "MxIO::save (sizes)");
      size_t tot = 0;
      for (int r = 0; r < np; ++r) tot += (size_t)counts[r] / sizeof(int64_t);
      allGids.resize(tot + 1);
      MX_ABI(mxg_comm_allgatherv(mComm->getCtx(), gids.data(), (int64_t)(nloc * sizeof(int64_t)), counts.data(), allGids.data()),
             "MxIO::save (ids)");
      allGids.resize(tot);
    }
    std::vector<Scalar> col, allVals;
    std::vector<double> re, im;
    if (me == 0) { re.resize(total); if (ScalarTraits<Scalar>::isComplex) im.resize(total); }
    for (size_t vec = 0; vec < fieldData.getNumVecs(); ++vec) {
      fieldData.getColumn(vec, col);
      const std::vector<int64_t>* ids = &gids;
      const std::vector<Scalar>* vals = &col;
      if (np > 1) {
        std::vector<int64_t> vcounts(np);
        for (int r = 0; r < np; ++r) vcounts[r] = counts[r] / (int64_t)sizeof(int64_t) * (int64_t)sizeof(Scalar);
        allVals.resize(allGids.size() + 1);
        MX_ABI(mxg_comm_allgatherv(mComm->getCtx(), col.data(), (int64_t)(nloc * sizeof(Scalar)), vcounts.data(), allVals.data()),
               "MxIO::save (values)");
        allVals.resize(allGids.size());
        ids = &allGids; vals = &allVals;
      }
      if (me != 0) continue;
      std::fill(re.begin(), re.end(), 0.0);
      std::fill(im.begin(), im.end(), 0.0);
      for (size_t l = 0; l < ids->size(); ++l) {
        const size_t g = (size_t)(*ids)[l];
        re[g] = mxRe((*vals)[l]);
        if (ScalarTraits<Scalar>::isComplex) im[g] = mxIm((*vals)[l]);
      }
      char stem[256], written[512];
      std::snprintf(stem, sizeof stem, "mx%s_%s_vec%.2i", field.getFieldName().c_str(), name.c_str(), (int)vec);
      MX_ABI_H(mxh_field_write(stem, "real", dims, re.data(), written, sizeof written), "MxIO::save");
      out.push_back(written);
      if (ScalarTraits<Scalar>::isComplex) {   // "imag" next to "real" in the same file (src/MxIO.cpp:199-222); a second .npy without HDF5
        MX_ABI_H(mxh_field_append(stem, "imag", dims, im.data(), written, sizeof written), "MxIO::save");
        if (out.back() != written) out.push_back(written);
      }
    }
    return out;
  }

 private:
  RCP<MxComm> mComm;
};

#endif
