// Eigen-driver (reference src/MxSolver.h:22-69, src/MxSolver.cpp:24-239): configures the
// (re-implemented) Anasazi block Krylov-Schur manager around MxMagWaveOp and post-processes.
#ifndef MX_SOLVER
#define MX_SOLVER

#include <iomanip>

#include "MxIO.h"
#include "MxMagWaveOp.h"

// inner product operator of the eigenproblem: x -> M x with M = mRhs (dmA).  The shift-invert
// operator P (L - sigma M)^-1 M is self-adjoint in the M inner product, which is what makes the
// Hermitian Krylov-Schur path legal; the reference leaves the Euclidean default in place
// (src/MxSolver.cpp:68-73), the converged eigenpairs are the same.
template <typename Scalar>
class MxMassOp : public Anasazi::Operator<Scalar> {
 public:
  explicit MxMassOp(RCP<MxCrsMatrix<Scalar> > m) : mM(m) {}
  virtual void Apply(Anasazi::MultiVec<Scalar> const& x, Anasazi::MultiVec<Scalar>& y) const {
    mM->apply(dynamic_cast<MxAnasaziMV<Scalar> const&>(x), dynamic_cast<MxAnasaziMV<Scalar>&>(y));
  }
 private:
  RCP<MxCrsMatrix<Scalar> > mM;
};

template <size_t DIM, typename Scalar>
class MxSolver {
 public:
  MxSolver(RCP<MxEMSim<DIM> > aSim, Teuchos::ParameterList aPList)
      : myPID(aSim->getGrid().getComm()->myPID()), sim(aSim), pList(aPList), mSolveSeconds(0) {
    sim->setParameters(pList);
    MxMagWaveOp<DIM, Scalar>* mwOp = new MxMagWaveOp<DIM, Scalar>(sim);
    op = rcp<OP>(mwOp);
    mwOpPtr = mwOp;
    std::string type = pList.get("eigensolver : type", "krylov-schur");
    int blockSize = pList.get("eigensolver : block size", 1);
    int numBlocks = pList.get("eigensolver : basis", 20);
    int maxRestarts = pList.get("eigensolver : max restarts", 20);
    double tol = pList.get("eigensolver : tol", 1.e-8);
    std::string targetSpectrum = pList.get("eigensolver : spectrum", "LM");
    int nev = pList.get("eigensolver : nev", 10);
    int verb = pList.get("eigensolver : output", 1);
    anasaziPList.set("Verbosity", verb);
    anasaziPList.set("Which", targetSpectrum);
    anasaziPList.set("Block Size", blockSize);
    anasaziPList.set("Num Blocks", numBlocks);
    anasaziPList.set("Maximum Restarts", maxRestarts);
    anasaziPList.set("Convergence Tolerance", tol);
    if (myPID == 0 && verb) std::cout << anasaziPList;

    RCP<MxAnasaziMV<Scalar> > initVec = rcp(new MxAnasaziMV<Scalar>(sim->getField("bfield").getMap(), blockSize));
    initVec->random();
    MxGridField<DIM>::zeroUnusedComponents(*initVec, sim->getField("bfield"));
    mInitVec = initVec;
    mEigProb = rcp(new Anasazi::BasicEigenproblem<Scalar, MV, OP>(op, mInitVec));
    mEigProb->setNEV(nev);
    // Hermitian path only for a pencil that really is symmetric: the reference calls setHermitian(true) whenever there
    // is no dielectric (src/MxSolver.cpp:68-73), but symmetry walls that mirror the on-wall component (pmc lower walls
    // of the octant problems, example/run.py) make the vector Laplacian non-symmetric at the wall, and a Hermitian
    // Krylov-Schur on it stagnates at residuals of 1e-7 (tests/test_tm_term_gpu.py found it on the empty PEC sphere)
    const bool hermitian = !sim->hasDielectric() && !sim->hasMu() && mwOp->isSelfAdjoint();
    if (hermitian) mEigProb->setHermitian(true);
    else if (!sim->hasDielectric() && !sim->hasMu() && myPID == 0 && verb)
      std::cout << "MxSolver: the operator is not symmetric (mirror boundary conditions): general Krylov-Schur\n";
    // The reference iterates in the Euclidean inner product (src/MxSolver.cpp:69-75 never calls setM).  With mRhs = dmA
    // (or I) - a real positive diagonal - the shift-inverted operator is self-adjoint in the mRhs inner product when
    // there are no materials, which is what the Hermitian Krylov-Schur path needs; that inner product is kept for the
    // dielectric case too.  With a <Mu> object mRhs = mu dmA is neither diagonal nor (with a PML) Hermitian, x^H mRhs y
    // is not an inner product any more, and the solver works in the Euclidean one as the reference does.
    if (!sim->hasMu()) mEigProb->setM(rcp<OP>(new MxMassOp<Scalar>(mwOp->getRhs())));
    if (!mEigProb->setProblem() && myPID == 0)
      std::cout << "Anasazi::BasicEigenproblem::setProblem() returned with error." << std::endl;
    if (type == "davidson" && hermitian && !ScalarTraits<Scalar>::isComplex) {
      // src/MxSolver.cpp:88-91: Anasazi's BlockDavidson with locking, on the same operator and without a preconditioner
      // of its own (the reference never calls setPrec): generalised Davidson expanding by the residuals
      anasaziPList.set("Use Locking", true);
      mEigMgr = rcp<Anasazi::SolverManager<Scalar, MV, OP> >(new Anasazi::BlockDavidsonSolMgr<Scalar, MV, OP>(mEigProb, anasaziPList));
    } else if (type == "krylov-schur" || type == "davidson") {
      // Anasazi's BlockDavidson takes Hermitian problems only (it throws otherwise): a dielectric / complex deck that
      // asks for it is served by the block Krylov-Schur manager, with a notice
      if (type == "davidson" && myPID == 0 && verb)
        std::cout << "MxSolver: eigensolver type 'davidson' needs a Hermitian problem -> block Krylov-Schur\n";
      mEigMgr = rcp<Anasazi::SolverManager<Scalar, MV, OP> >(new Anasazi::BlockKrylovSchurSolMgr<Scalar, MV, OP>(mEigProb, anasaziPList));
    } else {
      std::cout << "Unsupported eigensolver type: " << type << "\n";
      exit(EXIT_FAILURE);
    }
  }

  void solve() {   // src/MxSolver.cpp:100-239
    auto t0 = std::chrono::steady_clock::now();
    Anasazi::ReturnType returnCode = mEigMgr->solve();
    mSolveSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    mConverged = returnCode == Anasazi::Converged;
    const int verb = pList.get("eigensolver : output", 1);
    if (!mConverged && myPID == 0) std::cout << "Anasazi::EigensolverMgr::solve() returned unconverged." << std::endl;
    if (myPID == 0 && verb)
      std::cout << "\n\n---------------------------\n Eigensolve time: " << mSolveSeconds << " seconds\n---------------------------\n\n";
    Anasazi::Eigensolution<Scalar, MV> sol = mEigProb->getSolution();
    mBEigVecs = Teuchos::rcp_dynamic_cast<MxAnasaziMV<Scalar> >(sol.Evecs, true);
    mEigVals.resize(sol.numVecs);
    for (int k = 0; k < sol.numVecs; k++) mEigVals[k] = MxComplex(sol.Evals[k].realpart, sol.Evals[k].imagpart);
    mwOpPtr->checkEigensolution(mEigVals, *mBEigVecs, eigErrs);
    mwOpPtr->eigValsToFreqs(mEigVals, mEigFreqs);
    mwOpPtr->checkDivergences(*mBEigVecs, divErrs);
    if (myPID == 0 && verb) {
      std::cout << "\n========= Eigenfrequencies (Hz) =========\n\n";
      std::cout.setf(std::ios_base::left, std::ios_base::adjustfield);
      std::cout << std::setw(20) << "Real" << std::setw(20) << "Imag" << std::setw(20) << "Relative error" << "\n\n";
      std::cout.setf(std::ios_base::right, std::ios_base::adjustfield);
      std::cout.setf(std::ios::scientific);
      std::cout.precision(8);
      for (size_t i = 0; i < mEigFreqs.size(); ++i)
        std::cout << std::setw(3) << i << ":" << std::setw(16) << mEigFreqs[i].real() << std::setw(20)
                  << mEigFreqs[i].imag() << std::setw(20) << eigErrs[i] << "\n";
      std::cout << "\n=========================================\n";
      std::cout.unsetf(std::ios::scientific);
    }
    mEEigVecs = rcp(new MxMultiVector<Scalar>(sim->getField("efield").getMap(), sol.numVecs));
    mwOpPtr->magToElec(*mBEigVecs, *mEEigVecs);
    // src/MxSolver.cpp:222-224: the reference always writes the fields; here on request ("io : save fields",
    // set by the `maxwell` command line) - 3 (N+1)^3 doubles per vector is 0.4 GB at 256^3
    if (pList.get("io : save fields", false)) {
      MxIO<DIM, Scalar> io(sim->getGrid().getComm());
      std::vector<std::string> fb = io.save(*mBEigVecs, sim->getField("bfield"), "eigVecs");
      std::vector<std::string> fe = io.save(*mEEigVecs, sim->getField("efield"), "eigVecs");
      mFieldFiles = fb;
      mFieldFiles.insert(mFieldFiles.end(), fe.begin(), fe.end());
      if (myPID == 0 && verb) for (auto const& f : mFieldFiles) std::cout << "Saved " << f << "\n";
    }
  }

  // results (the reference only prints them)
  bool converged() const { return mConverged; }
  std::vector<MxComplex> const& eigenvalues() const { return mEigVals; }
  std::vector<MxComplex> const& frequencies() const { return mEigFreqs; }
  std::vector<double> const& residuals() const { return eigErrs; }
  std::vector<double> const& divergences() const { return divErrs; }
  RCP<MxMultiVector<Scalar> > magneticEigenvectors() const { return mBEigVecs; }
  RCP<MxMultiVector<Scalar> > electricEigenvectors() const { return mEEigVecs; }
  double eigensolveSeconds() const { return mSolveSeconds; }
  std::vector<std::string> const& fieldFiles() const { return mFieldFiles; }
  MxMagWaveOp<DIM, Scalar> const& waveOp() const { return *mwOpPtr; }

 private:
  typedef Anasazi::Operator<Scalar> OP;
  typedef Anasazi::MultiVec<Scalar> MV;
  int myPID;
  RCP<MxEMSim<DIM> > sim;
  Teuchos::ParameterList pList, anasaziPList;
  RCP<OP> op;
  MxMagWaveOp<DIM, Scalar>* mwOpPtr;
  RCP<MV> mInitVec;
  std::vector<MxComplex> mEigVals, mEigFreqs;
  std::vector<double> eigErrs, divErrs;
  std::vector<std::string> mFieldFiles;
  RCP<MxMultiVector<Scalar> > mBEigVecs, mEEigVecs;
  RCP<Anasazi::BasicEigenproblem<Scalar, MV, OP> > mEigProb;
  RCP<Anasazi::SolverManager<Scalar, MV, OP> > mEigMgr;
  double mSolveSeconds;
  bool mConverged = false;
};

#endif
