// Internal definitions shared by the mxh_* C ABI translation units.
#ifndef MXH_INTERNAL_HPP
#define MXH_INTERNAL_HPP

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "mx_yee.hpp"
#include "mxh.h"

struct mxh_sim {
  mx::YeeConfig cfg;
  std::shared_ptr<mx::YeeAssembler<double>> asmb;
  std::map<std::string, mx::Csr<double>> ops;
  std::map<double, mx::Csr<double>> shifted;
  std::vector<double> mask;
  double setup_seconds = 0.0;
  bool chain_ready = true;   // false: only the single operators are in `ops` yet (MXH_DEVICE_CHAIN), see mxh_sim_ensure_chain
  // distributed exports (kept alive for the borrowed pointers)
  struct DistOp {
    mx::UVec<int64_t> rowptr, colgid, rowgid, colowned;
    mx::UVec<double> vals;
  };
  std::map<std::string, DistOp> dist;
};

void mxh_set_error(const std::string& s);
std::shared_ptr<const mx::Shape> mxh_parse_shape(const char* spec);
const mx::Csr<double>& mxh_sim_shifted_matrix(mxh_sim* s, double shift);
void mxh_sim_ensure_chain(mxh_sim* s);
// the operator chain formed on the GPU of `ctx` from the single operators in `ops` (mxh_sim.cpp); S = double or std::complex<double>
template <typename S>
void mxh_device_chain(mxg_ctx* ctx, const mx::YeeAssembler<S>& a, std::map<std::string, mx::Csr<S> >& ops);

#endif
