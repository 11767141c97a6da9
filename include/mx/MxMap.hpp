// Row-index set owned by this rank (GID <-> LID).  Reference src/MxMap.hpp:17-143.
#ifndef MX_MAP
#define MX_MAP

#include <algorithm>
#include <functional>
#include <unordered_map>

#include "MxComm.hpp"

class MxMap {
 public:
  // linear map with numGlobal indices split evenly (serial: all local); src/MxMap.hpp:22
  MxMap(size_t numGlobal, RCP<MxComm> comm) : comm_(comm), numGlobal_(numGlobal) {
    const size_t p = comm->numProc(), r = comm->myPID();
    const size_t lo = numGlobal * r / p, hi = numGlobal * (r + 1) / p;
    for (size_t i = lo; i < hi; ++i) gids_.push_back(i);
  }
  // arbitrary local index list; numGlobal = sum over ranks (serial: the local count)
  MxMap(const std::vector<MxIndex>& globalIndices, RCP<MxComm> comm)
      : gids_(globalIndices), comm_(comm), numGlobal_(globalIndices.size()) { buildLookup(); }
  // the same, taking the list over (the 2e7-entry maps of a 256^3 problem are not copied)
  MxMap(std::vector<MxIndex>&& globalIndices, RCP<MxComm> comm)
      : gids_(std::move(globalIndices)), comm_(comm), numGlobal_(gids_.size()) { buildLookup(); }
  MxMap(size_t numGlobal, std::vector<MxIndex>&& globalIndices, RCP<MxComm> comm)
      : gids_(std::move(globalIndices)), comm_(comm), numGlobal_(numGlobal) { buildLookup(); }
  MxMap(size_t numGlobal, const std::vector<MxIndex>& globalIndices, RCP<MxComm> comm)
      : gids_(globalIndices), comm_(comm), numGlobal_(numGlobal) { buildLookup(); }

  size_t getNodeNumIndices() const { return gids_.size(); }
  size_t getGlobalNumIndices() const { return numGlobal_; }
  void setGlobalNumIndices(size_t n) { numGlobal_ = n; }
  MxIndex getGlobalIndex(MxIndex lid) const { return gids_[lid]; }
  // returns size_t(-1) when the index is not on this rank (Teuchos::OrdinalTraits::invalid())
  MxIndex getLocalIndex(MxIndex gid) const {
    if (sorted_) {
      auto it = std::lower_bound(gids_.begin(), gids_.end(), gid);
      return (it != gids_.end() && *it == gid) ? MxIndex(it - gids_.begin()) : MxIndex(-1);
    }
    auto it = lookup_.find(gid);
    return it == lookup_.end() ? MxIndex(-1) : it->second;
  }
  bool isNodeGlobalIndex(MxIndex gid) const { return getLocalIndex(gid) != MxIndex(-1); }
  MxIndex const* getNodeIndexList() const { return gids_.data(); }
  const std::vector<MxIndex>& getNodeIndices() const { return gids_; }
  RCP<MxComm> getComm() const { return comm_; }
  // rank owning a global index that is not local (the directory Epetra builds with MPI; here the grid
  // partition answers it, src/MxGrid.cpp:171-193); set by MxEMSim for the maps of a decomposed grid
  void setOwnerFunction(std::function<int(MxIndex)> f) { owner_ = f; }
  bool hasOwnerFunction() const { return (bool)owner_; }
  int ownerOf(MxIndex gid) const { return owner_(gid); }

 private:
  void buildLookup() {
    sorted_ = std::is_sorted(gids_.begin(), gids_.end());
    if (!sorted_) for (size_t i = 0; i < gids_.size(); ++i) lookup_[gids_[i]] = i;
  }
  std::vector<MxIndex> gids_;
  RCP<MxComm> comm_;
  size_t numGlobal_;
  bool sorted_ = true;
  std::unordered_map<MxIndex, MxIndex> lookup_;
  std::function<int(MxIndex)> owner_;
};

#endif
