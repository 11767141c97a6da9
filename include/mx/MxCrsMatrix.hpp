// Sparse matrix with the reference's interface (src/MxCrsMatrix.hpp:11-104).  Rows are collected
// on the host (insertRowValues), fillComplete() freezes them into CSR and uploads to the GPU;
// apply() is the HIP SpMV / SpMM of mxg_csr_apply.
#ifndef MX_CRS_MATRIX
#define MX_CRS_MATRIX

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "MxVector.hpp"
#include "mx_sparse.hpp"

template <typename Scalar>
class MxCrsMatrix {
 public:
  explicit MxCrsMatrix(RCP<MxMap> rowMap) : mFilled(false), mRowMap(rowMap), mColMap(rowMap), mDev(nullptr) { initRows(); }
  MxCrsMatrix(RCP<MxMap> rowMap, RCP<MxMap> colMap) : mFilled(false), mRowMap(rowMap), mColMap(colMap), mDev(nullptr) { initRows(); }
  // diagonal matrix from a scalar (src/MxCrsMatrix.cpp:33-47)
  MxCrsMatrix(RCP<MxMap> rowMap, Scalar diag) : mFilled(false), mRowMap(rowMap), mColMap(rowMap), mDev(nullptr) {
    initRows();
    for (size_t i = 0; i < rowMap->getNodeNumIndices(); ++i) {
      MxIndex g = rowMap->getGlobalIndex(i);
      insertRowValues(g, 1, &g, &diag);
    }
    fillComplete(rowMap, rowMap);
  }
  // diagonal matrix from a vector (src/MxCrsMatrix.cpp:50-66)
  explicit MxCrsMatrix(RCP<MxVector<Scalar> > diag) : mFilled(false), mRowMap(diag->getMap()), mColMap(diag->getMap()), mDev(nullptr) {
    initRows();
    std::vector<Scalar> d;
    diag->getColumn(0, d);
    for (size_t i = 0; i < d.size(); ++i) {
      MxIndex g = mRowMap->getGlobalIndex(i);
      insertRowValues(g, 1, &g, &d[i]);
    }
    fillComplete(mRowMap, mRowMap);
  }
  // adopt a finished local CSR (the fast path MxEMOps uses)
  MxCrsMatrix(RCP<MxMap> rowMap, RCP<MxMap> colMap, mx::Csr<Scalar> const& csr)
      : mFilled(false), mRowMap(rowMap), mColMap(colMap), mDev(nullptr) { adopt(csr); }
  // ... without copying it: the arrays stay the caller's (the operators a sim has assembled outlive the solver built on
  // them).  The device copy is made from them; a host copy only if something modifies or reads the matrix on the
  // host.  At 256^3 the single operators are 3.3 GB: copying them was 1 s of the setup and 3 GB of resident memory.
  MxCrsMatrix(RCP<MxMap> rowMap, RCP<MxMap> colMap, mx::Csr<Scalar> const* borrowed)
      : mFilled(true), mRowMap(rowMap), mColMap(colMap), mDev(nullptr), mBorrow(borrowed) { mHostValid = false; }

  // Row-partitioned operator of a decomposed grid: the owned rows with GLOBAL column ids.  Columns this rank
  // does not own become ghosts and the import plan is agreed with the other ranks here, which is what Epetra's
  // FillComplete does (src/MxCrsMatrix.cpp:325-342).  COLLECTIVE: every rank constructs its part in the same order.
  MxCrsMatrix(RCP<MxMap> rowMap, RCP<MxMap> domainMap, mx::UVec<int64_t> const& rowptr,
              mx::UVec<int64_t> const& colGids, mx::UVec<Scalar> const& vals)
      : mFilled(false), mRowMap(rowMap), mColMap(domainMap), mDev(nullptr) { adoptDistributed(rowptr, colGids, vals); }

  ~MxCrsMatrix() { if (mDev) mxg_csr_destroy(mDev); }

  // indices are always global here
  void insertRowValues(MxIndex row, std::vector<MxIndex> cols, std::vector<Scalar> vals) {
    insertRowValues(row, cols.size(), cols.data(), vals.data());
  }
  void insertRowValues(MxIndex row, size_t numEntries, MxIndex const* cols, Scalar const* vals) {
    if (mFilled) { std::cout << "MxCrsMatrix::insertRowValues: matrix already filled\n"; throw 1; }
    const MxIndex lr = mRowMap->getLocalIndex(row);
    if (lr == MxIndex(-1)) {
      std::cout << "MxCrsMatrix::insertRowValues: row=" << row << " is not on this rank. numEntries=" << numEntries << "\n";
      throw 1;
    }
    needRows();
    for (size_t i = 0; i < numEntries; ++i) mRows[lr].push_back(std::make_pair(cols[i], vals[i]));
  }

  void fillComplete(RCP<MxMap> domainMap, RCP<MxMap> rangeMap) {
    mColMap = domainMap;
    (void)rangeMap;
    mx::Csr<Scalar> m;
    m.nrows = (int64_t)mRowMap->getNodeNumIndices();
    m.ncols = (int64_t)domainMap->getNodeNumIndices();
    m.rowptr.resize(m.nrows + 1);
    m.rowptr[0] = 0;
    std::vector<std::pair<int32_t, Scalar> > row;
    std::vector<int32_t> col; std::vector<Scalar> val;
    needRows();
    for (int64_t r = 0; r < m.nrows; ++r) {
      row.clear();
      for (auto& e : mRows[r]) {
        const MxIndex lc = domainMap->getLocalIndex(e.first);
        if (lc == MxIndex(-1)) { std::cout << "MxCrsMatrix::fillComplete: column " << e.first << " not in the domain map\n"; throw 1; }
        row.push_back(std::make_pair((int32_t)lc, e.second));
      }
      std::stable_sort(row.begin(), row.end(), [](const std::pair<int32_t, Scalar>& a, const std::pair<int32_t, Scalar>& b) { return a.first < b.first; });
      for (size_t i = 0; i < row.size(); ++i) {   // repeated (row, col) inserts are summed, like Epetra
        if (!col.empty() && (int64_t)col.size() > m.rowptr[r] && col.back() == row[i].first) val.back() += row[i].second;
        else { col.push_back(row[i].first); val.push_back(row[i].second); }
      }
      m.rowptr[r + 1] = (int64_t)col.size();
    }
    m.col.assign(col.begin(), col.end());
    m.val.assign(val.begin(), val.end());
    mRows.clear();
    adopt(m);
  }

  void scale(Scalar val) { host(); mx::scaleInPlace(mHost, val); upload(); }
  void leftScale(RCP<MxVector<Scalar> > vec) {
    std::vector<Scalar> d;
    vec->getColumn(0, d);
    mx::UVec<Scalar> dd(d.begin(), d.end());
    host();
    for (int64_t i = 0; i < mHost.nrows; ++i)
      for (int64_t p = mHost.rowptr[i]; p < mHost.rowptr[i + 1]; ++p) mHost.val[p] *= dd[i];
    upload();
  }
  void getLocalRow(MxIndex localRowIndex, std::vector<MxIndex>& localColIndices, std::vector<Scalar>& vals) const {
    localColIndices.clear(); vals.clear();
    host();
    for (int64_t p = mHost.rowptr[localRowIndex]; p < mHost.rowptr[localRowIndex + 1]; ++p) {
      localColIndices.push_back((MxIndex)mHost.col[p]);
      vals.push_back(mHost.val[p]);
    }
  }
  RCP<MxVector<Scalar> > getRowSums(bool inverse) const {
    RCP<MxVector<Scalar> > res = rcp(new MxVector<Scalar>(mRowMap));
    host();
    std::vector<Scalar> s(mHost.nrows);
    for (int64_t i = 0; i < mHost.nrows; ++i) {
      Scalar sum = Scalar(0);   // the reference never resets `sum` (src/MxCrsMatrix.cpp:306-311)
      for (int64_t p = mHost.rowptr[i]; p < mHost.rowptr[i + 1]; ++p) sum += mHost.val[p];
      s[i] = inverse ? Scalar(1.0) / sum : sum;
    }
    res->setColumn(0, s);
    return res;
  }

  // y = A x on the GPU; x may alias y (src/MxMagWaveOp.cpp:650,1249)
  void apply(MxMultiVector<Scalar> const& x, MxMultiVector<Scalar>& y) const {
    if (!mFilled) { std::cout << "MxCrsMatrix::apply: matrix is not fill-completed\n"; throw 1; }
    // re-encoding costs a host-side pass over the matrix: operators the hot loop uses are encoded at setup
    // (getRawMatrix), anything else once it has been applied a few times - the one-off products of the
    // post-processing (curlCurl, curlB in checkEigensolution / magToElec) stay plain CSR
    ensureDevice(mApplies.fetch_add(1) >= 3);
    MX_ABI(mxg_csr_apply(mRowMap->getComm()->getCtx(), mDev, x.raw(), y.raw()), "MxCrsMatrix::apply");
  }

  // MatrixMarket coordinate file <name>.mm (EpetraExt::RowMatrixToMatrixMarketFile)
  void save(std::string name) const {
    std::ofstream f((name + ".mm").c_str());
    f.precision(17);
    host();
    f << "%%MatrixMarket matrix coordinate " << (ScalarTraits<Scalar>::isComplex ? "complex" : "real") << " general\n";
    f << mRowMap->getGlobalNumIndices() << " " << mColMap->getGlobalNumIndices() << " " << mHost.nnz() << "\n";
    for (int64_t i = 0; i < mHost.nrows; ++i)
      for (int64_t p = mHost.rowptr[i]; p < mHost.rowptr[i + 1]; ++p) {
        f << mRowMap->getGlobalIndex(i) + 1 << " " << mColMap->getGlobalIndex(mHost.col[p]) + 1 << " " << mxRe(mHost.val[p]);
        if (ScalarTraits<Scalar>::isComplex) f << " " << mxIm(mHost.val[p]);
        f << "\n";
      }
  }

  void purgeZeros() {   // |a| > 1e-12 kept
    if (mDev && !mHostValid) {          // born on the device: purge there (mxg_csr_purge, same rule)
      mxg_csr* r = nullptr;
      MX_ABI(mxg_csr_purge(mRowMap->getComm()->getCtx(), mDev, 1.e-12, &r), "MxCrsMatrix::purgeZeros");
      mxg_csr_destroy(mDev);
      mDev = r;
      mBorrow = nullptr;                 // the content changed: the borrowed arrays are not this matrix any more
      mCompressed = false;
      return;
    }
    mx::Csr<Scalar> p = mx::purge(host());
    adopt(p);
  }

  void fillComplete() { fillComplete(mColMap, mRowMap); }
  bool isFilled() const { return mFilled; }
  RCP<MxMap> getDomainMap() const { return mColMap; }
  RCP<MxMap> getRangeMap() const { return mRowMap; }

  static void matrixMatrixMultiply(MxCrsMatrix<Scalar> const& a, bool transA, MxCrsMatrix<Scalar> const& b,
                                   bool transB, MxCrsMatrix<Scalar>& result, bool fc) {
    result.mRowMap = transA ? a.mColMap : a.mRowMap;
    result.mColMap = transB ? b.mRowMap : b.mColMap;
    (void)fc;
    if (!transA && !transB && deviceAlgebra(a, b)) {
      // the product is formed on the GPU (mxg_csr_multiply: the arithmetic and order of mx::multiply, bit for bit);
      // a host copy is only made if somebody asks for one (getHostCsr, getLocalRow, save)
      a.ensureDevice(false); b.ensureDevice(false);
      mxg_csr* r = nullptr;
      MX_ABI(mxg_csr_multiply(a.mRowMap->getComm()->getCtx(), a.mDev, b.mDev, &r), "MxCrsMatrix::matrixMatrixMultiply");
      result.adoptDevice(r);
      return;
    }
    mx::Csr<Scalar> ta, tb;
    const mx::Csr<Scalar>& A = transA ? (ta = mx::transpose(a.host(), true)) : a.host();
    const mx::Csr<Scalar>& B = transB ? (tb = mx::transpose(b.host(), true)) : b.host();
    mx::Csr<Scalar> c = mx::multiply(A, B);
    result.adopt(c);
  }
  static void matrixMatrixAdd(MxCrsMatrix<Scalar> const& a, bool transA, Scalar scalarA,
                              MxCrsMatrix<Scalar> const& b, bool transB, Scalar scalarB,
                              MxCrsMatrix<Scalar>& result, bool fc) {
    result.mRowMap = transA ? a.mColMap : a.mRowMap;
    result.mColMap = transA ? a.mRowMap : a.mColMap;
    (void)fc;
    if (!transA && !transB && deviceAlgebra(a, b)) {
      a.ensureDevice(false); b.ensureDevice(false);
      mxg_csr* r = nullptr;
      const double za[2] = {mxRe(scalarA), mxIm(scalarA)}, zb[2] = {mxRe(scalarB), mxIm(scalarB)};
      MX_ABI(mxg_csr_add_z(a.mRowMap->getComm()->getCtx(), a.mDev, za, b.mDev, zb, &r), "MxCrsMatrix::matrixMatrixAdd");
      result.adoptDevice(r);
      return;
    }
    mx::Csr<Scalar> ta, tb;
    const mx::Csr<Scalar>& A = transA ? (ta = mx::transpose(a.host(), true)) : a.host();
    const mx::Csr<Scalar>& B = transB ? (tb = mx::transpose(b.host(), true)) : b.host();
    mx::Csr<Scalar> c = mx::add(A, scalarA, B, scalarB);
    result.adopt(c);
  }

  // host-side CSR (setup algebra, tests) and device handle (advanced usage)
  mx::Csr<Scalar> const& getHostCsr() const { return host(); }
  mxg_csr* getRawMatrix() const { ensureDevice(); return mDev; }
  // give the host copy back once the device holds the matrix (it returns on demand: host()); the 256^3 shifted operator
  // is 2.2 GB that nothing reads again after the multigrid setup
  void releaseHostCopy() const {
    if (mDev && mHostValid && !mBorrow) { mHost = mx::Csr<Scalar>(); mHostValid = false; }
  }
  // never re-encode this matrix (operators outside the inner loops whose encoding pass would cost more than it saves)
  void keepPlain() const { mNoCompress = true; }
  // the setup-time products / sums of single-rank matrices (real or complex) run on the GPU (MXH_DEVICE_CHAIN=0: on the host)
  static bool deviceChainEnabled() {
    static const bool on = !(getenv("MXH_DEVICE_CHAIN") && atoi(getenv("MXH_DEVICE_CHAIN")) == 0);
    return on;
  }

 private:
  // the per-row insertion lists exist only while somebody inserts: a result matrix of the operator chain (formed by
  // matrixMatrixMultiply / Add, never inserted into) used to allocate and zero 20.6 M empty lists (0.5 GB) at 256^3
  void initRows() {}
  void needRows() { if (mRows.size() != (size_t)mRowMap->getNodeNumIndices()) mRows.resize(mRowMap->getNodeNumIndices()); }
  // the device copy is created on first use: setup-only intermediates never reach the GPU
  void adopt(mx::Csr<Scalar> const& m) { mHost = m; mHostValid = true; mBorrow = nullptr; upload(); }
  void upload() {
    if (mDev) { mxg_csr_destroy(mDev); mDev = nullptr; }
    mCompressed = false;
    mFilled = true;
  }
  // a matrix formed on the device: the handle is the matrix, the host copy follows on demand
  void adoptDevice(mxg_csr* r) {
    if (mDev) mxg_csr_destroy(mDev);
    mDev = r;
    mHost = mx::Csr<Scalar>();
    mHostValid = false;
    mBorrow = nullptr;
    mCompressed = false;
    mFilled = true;
    mRows.clear();
  }
  static bool deviceAlgebra(MxCrsMatrix<Scalar> const& a, MxCrsMatrix<Scalar> const& b) {
    return deviceChainEnabled() && !a.mDistributed && !b.mDistributed && a.mRowMap->getComm()->numProc() == 1;
  }
  mx::Csr<Scalar> const& host() const {
    if (!mHostValid && mBorrow) {                 // borrowed arrays: the host copy is made when somebody needs one
      mHost = *mBorrow;
      mHostValid = true;
    }
    if (!mHostValid) {
      int64_t nr = 0, nc = 0, ng = 0, nz = 0;
      int cx = 0;
      MX_ABI(mxg_csr_info(mDev, &nr, &nc, &ng, &nz, &cx), "MxCrsMatrix (host copy)");
      mHost.nrows = nr; mHost.ncols = nc + ng;
      mHost.rowptr.resize(nr + 1);
      mHost.col.resize(nz);
      mHost.val.resize(nz);
      MX_ABI(mxg_csr_download(mDev, mHost.rowptr.data(), mHost.col.data(), reinterpret_cast<double*>(mHost.val.data())),
             "MxCrsMatrix (host copy)");
      mHostValid = true;
    }
    return mHost;
  }
  void ensureDevice(bool compressIt = true) const {
    static const bool compressEnv = !(getenv("MXG_COMPRESS") && atoi(getenv("MXG_COMPRESS")) == 0);
    const bool compress = compressEnv && !mNoCompress;
    if (mDev) {
      if (compressIt && compress && !mCompressed && mRowMap->getNodeNumIndices() >= 256) {
        // a host copy at hand (the multigrid setup has read the matrix back) saves the library its own read-back
        if (mHostValid)
          MX_ABI(mxg_csr_compress_from(mDev, mHost.rowptr.data(), mHost.col.data(), reinterpret_cast<const double*>(mHost.val.data()),
                                       NULL, NULL, NULL), "MxCrsMatrix (compress)");
        else
          MX_ABI(mxg_csr_compress(mDev, NULL, NULL, NULL), "MxCrsMatrix (compress)");
        mCompressed = true;
      }
      return;
    }
    mxg_halo_plan plan;
    plan.nneigh = (int)mNeigh.size();
    plan.neigh = mNeigh.data();
    plan.send_off = mSendOff.data();
    plan.send_idx = mSendIdx.empty() ? &mNoIdx : mSendIdx.data();
    plan.recv_off = mRecvOff.data();
    const bool halo = mDistributed && !mNeigh.empty();
    const mx::Csr<Scalar>& H = (!mHostValid && mBorrow) ? *mBorrow : mHost;
    MX_ABI(mxg_csr_create(mRowMap->getComm()->getCtx(), H.nrows, H.ncols - mNumGhost, mNumGhost, H.rowptr.data(),
                          H.col.data(), reinterpret_cast<const double*>(H.val.data()),
                          ScalarTraits<Scalar>::isComplex, halo ? &plan : NULL, &mDev), "MxCrsMatrix::fillComplete");
    // lattice operators (real or complex) take the tile form - no matrix stream for the bulk rows - other real
    // matrices the compressed stream (mxg_csr_compress; bit-identical results); MXG_COMPRESS=0 keeps the plain CSR
    // kernels.  With ghost columns the tiles and the rows that reference none run while the halo is in flight.
    if (compressIt && compress && H.nrows >= 256) {
      MX_ABI(mxg_csr_compress_from(mDev, H.rowptr.data(), H.col.data(),
                                   reinterpret_cast<const double*>(H.val.data()), NULL, NULL, NULL),
             "MxCrsMatrix::fillComplete (compress)");
      mCompressed = true;
    }
  }
  // ghosts = the global column ids this rank does not own, grouped by owner (ascending rank, then ascending id);
  // local column numbering [owned | ghosts].  One all-gather of the need lists gives every rank its send side.
  void adoptDistributed(mx::UVec<int64_t> const& rowptr, mx::UVec<int64_t> const& colGids, mx::UVec<Scalar> const& vals) {
    RCP<MxComm> comm = mRowMap->getComm();
    const int me = comm->myPID(), np = comm->numProc();
    const int64_t nowned = (int64_t)mColMap->getNodeNumIndices(), nnz = (int64_t)colGids.size();
    mx::UVec<int64_t> lc(nnz);
    std::vector<std::pair<int64_t, int64_t> > need;   // (owner, gid)
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < nnz; ++q) {
      const MxIndex l = mColMap->getLocalIndex((MxIndex)colGids[q]);
      lc[q] = l == MxIndex(-1) ? -1 : (int64_t)l;
    }
    for (int64_t q = 0; q < nnz; ++q) if (lc[q] < 0) need.push_back(std::make_pair((int64_t)0, colGids[q]));
    if (!need.empty() && (np == 1 || !mColMap->hasOwnerFunction())) {
      std::cout << "MxCrsMatrix: column " << need[0].second << " is not in the domain map of this rank and the map has no owner directory\n";
      throw 1;
    }
    for (auto& e : need) e.first = mColMap->ownerOf((MxIndex)e.second);
    std::sort(need.begin(), need.end());
    need.erase(std::unique(need.begin(), need.end()), need.end());
    mNumGhost = (int64_t)need.size();
    mGhostGids.resize(need.size());
    for (size_t i = 0; i < need.size(); ++i) mGhostGids[i] = need[i].second;
    for (int64_t q = 0; q < nnz; ++q)
      if (lc[q] < 0) {
        const std::pair<int64_t, int64_t> key((int64_t)mColMap->ownerOf((MxIndex)colGids[q]), colGids[q]);
        lc[q] = nowned + (std::lower_bound(need.begin(), need.end(), key) - need.begin());
      }
    mHost.nrows = (int64_t)mRowMap->getNodeNumIndices();
    mHost.ncols = nowned + mNumGhost;
    if (mHost.ncols >= ((int64_t)1 << 31)) { std::cout << "MxCrsMatrix: more than 2^31 local columns\n"; throw 1; }
    mHost.rowptr.assign(rowptr.begin(), rowptr.end());
    mHost.col.resize(nnz);
    for (int64_t q = 0; q < nnz; ++q) mHost.col[q] = (int32_t)lc[q];
    mHost.val.assign(vals.begin(), vals.end());
    mDistributed = true;
    // plan agreement
    mx::UVec<int64_t> flat(2 * need.size());
    for (size_t i = 0; i < need.size(); ++i) { flat[2 * i] = need[i].first; flat[2 * i + 1] = need[i].second; }
    std::vector<int64_t> counts(np, 0);
    MX_ABI(mxg_comm_allgather_i64(comm->getCtx(), (int64_t)(flat.size() * sizeof(int64_t)), counts.data()), "MxCrsMatrix (plan sizes)");
    int64_t total = 0;
    std::vector<int64_t> off(np + 1, 0);
    for (int r = 0; r < np; ++r) { off[r + 1] = off[r] + counts[r] / (int64_t)sizeof(int64_t); total += counts[r]; }
    mx::UVec<int64_t> all((size_t)(total / (int64_t)sizeof(int64_t)) + 1);
    MX_ABI(mxg_comm_allgatherv(comm->getCtx(), flat.data(), (int64_t)(flat.size() * sizeof(int64_t)), counts.data(), all.data()),
           "MxCrsMatrix (plan exchange)");
    std::vector<std::vector<int64_t> > wanted(np);    // owned local indices rank r asks of me, in r's ghost order
    for (int r = 0; r < np; ++r) {
      if (r == me) continue;
      for (int64_t i = off[r]; i < off[r + 1]; i += 2) {
        if (all[i] != me) continue;
        const MxIndex l = mColMap->getLocalIndex((MxIndex)all[i + 1]);
        if (l == MxIndex(-1)) { std::cout << "MxCrsMatrix: rank " << r << " asks rank " << me << " for entry " << all[i + 1] << ", which it does not own\n"; throw 1; }
        wanted[r].push_back((int64_t)l);
      }
    }
    mNeigh.clear(); mSendOff.assign(1, 0); mRecvOff.assign(1, 0); mSendIdx.clear();
    size_t g = 0;
    for (int r = 0; r < np; ++r) {
      int64_t nrecv = 0;
      while (g < need.size() && need[g].first == r) { ++g; ++nrecv; }
      if (nrecv == 0 && wanted[r].empty()) continue;
      mNeigh.push_back(r);
      mSendIdx.insert(mSendIdx.end(), wanted[r].begin(), wanted[r].end());
      mSendOff.push_back((int64_t)mSendIdx.size());
      mRecvOff.push_back(mRecvOff.back() + nrecv);
    }
    mFilled = true;
  }

 public:
  bool isDistributed() const { return mDistributed; }
  int64_t numGhosts() const { return mNumGhost; }
  // global ids of the ghost columns, in local column order (local column = owned count + position)
  std::vector<int64_t> const& ghostGids() const { return mGhostGids; }
  // Setup-time ghost import of host data along the matrix' plan: one value per owned column in, one per ghost
  // column out.  COLLECTIVE (an all-gather of the packed send sides; the device path is mxg_csr_apply's exchange).
  template <typename T>
  std::vector<T> importGhostsHost(std::vector<T> const& owned) const {
    RCP<MxComm> comm = mRowMap->getComm();
    const int me = comm->myPID(), np = comm->numProc();
    std::vector<T> ghosts((size_t)mNumGhost);
    if (np == 1) return ghosts;
    // message: [nneigh, (peer, count) x nneigh] as int64, then the packed values
    const size_t nn = mNeigh.size();
    std::vector<char> msg(sizeof(int64_t) * (1 + 2 * nn) + sizeof(T) * mSendIdx.size());
    int64_t* head = reinterpret_cast<int64_t*>(msg.data());
    head[0] = (int64_t)nn;
    for (size_t i = 0; i < nn; ++i) { head[1 + 2 * i] = mNeigh[i]; head[2 + 2 * i] = mSendOff[i + 1] - mSendOff[i]; }
    T* pay = reinterpret_cast<T*>(msg.data() + sizeof(int64_t) * (1 + 2 * nn));
    for (size_t i = 0; i < mSendIdx.size(); ++i) pay[i] = owned[(size_t)mSendIdx[i]];
    std::vector<int64_t> counts(np, 0);
    MX_ABI(mxg_comm_allgather_i64(comm->getCtx(), (int64_t)msg.size(), counts.data()), "MxCrsMatrix::importGhostsHost (sizes)");
    int64_t total = 0;
    std::vector<int64_t> off(np + 1, 0);
    for (int r = 0; r < np; ++r) { off[r + 1] = off[r] + counts[r]; total += counts[r]; }
    std::vector<char> all((size_t)total + 8);
    MX_ABI(mxg_comm_allgatherv(comm->getCtx(), msg.data(), (int64_t)msg.size(), counts.data(), all.data()),
           "MxCrsMatrix::importGhostsHost (exchange)");
    for (size_t i = 0; i < nn; ++i) {
      const int p = mNeigh[i];
      const int64_t nrecv = mRecvOff[i + 1] - mRecvOff[i];
      if (nrecv == 0) continue;
      const char* m = all.data() + off[p];
      int64_t h0;
      std::memcpy(&h0, m, sizeof(int64_t));
      int64_t skip = 0, cnt = -1;
      for (int64_t j = 0; j < h0; ++j) {
        int64_t peer, c;
        std::memcpy(&peer, m + sizeof(int64_t) * (1 + 2 * j), sizeof(int64_t));
        std::memcpy(&c, m + sizeof(int64_t) * (2 + 2 * j), sizeof(int64_t));
        if (peer == me) { cnt = c; break; }
        skip += c;
      }
      if (cnt != nrecv) { std::cout << "MxCrsMatrix::importGhostsHost: rank " << p << " sends " << cnt << " values, rank " << me << " expects " << nrecv << "\n"; throw 1; }
      std::memcpy(ghosts.data() + mRecvOff[i], m + sizeof(int64_t) * (1 + 2 * h0) + sizeof(T) * skip, sizeof(T) * (size_t)nrecv);
    }
    return ghosts;
  }
  // square part on the owned columns (the block a rank-local preconditioner works on)
  mx::Csr<Scalar> ownedBlock() const {
    mx::Csr<Scalar> b;
    const int64_t nowned = mHost.ncols - mNumGhost;
    b.nrows = mHost.nrows; b.ncols = nowned;
    b.rowptr.resize(b.nrows + 1);
    b.rowptr[0] = 0;
    for (int64_t i = 0; i < mHost.nrows; ++i) {
      int64_t c = 0;
      for (int64_t q = mHost.rowptr[i]; q < mHost.rowptr[i + 1]; ++q) c += mHost.col[q] < nowned;
      b.rowptr[i + 1] = b.rowptr[i] + c;
    }
    b.col.resize(b.rowptr[b.nrows]); b.val.resize(b.rowptr[b.nrows]);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < mHost.nrows; ++i) {
      int64_t o = b.rowptr[i];
      for (int64_t q = mHost.rowptr[i]; q < mHost.rowptr[i + 1]; ++q)
        if (mHost.col[q] < nowned) { b.col[o] = mHost.col[q]; b.val[o] = mHost.val[q]; ++o; }
    }
    return b;
  }

 private:
  bool mDistributed = false;
  int64_t mNumGhost = 0;
  int64_t mNoIdx = 0;
  std::vector<int> mNeigh;
  std::vector<int64_t> mSendOff, mSendIdx, mRecvOff;
  std::vector<int64_t> mGhostGids;
  bool mFilled;
  RCP<MxMap> mRowMap, mColMap;
  std::vector<std::vector<std::pair<MxIndex, Scalar> > > mRows;
  mutable mx::Csr<Scalar> mHost;     // (filled on demand for a matrix formed on the device: host())
  mx::Csr<Scalar> const* mBorrow = nullptr;   // the caller's arrays of a borrowed matrix (valid until the matrix is modified)
  mutable bool mHostValid = true;
  mutable bool mCompressed = false, mNoCompress = false;
  mutable std::atomic<int> mApplies{0};    // applies so far (a matrix may be applied from several rank threads)
  mutable mxg_csr* mDev;
};

#endif
