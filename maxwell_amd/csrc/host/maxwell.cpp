// `maxwell --infile=your_file.mx`: the reference's command line (src/maxwell.cpp:41-120) on one
// MI355X.  Reads the deck, builds MxProblem<3>, solves and prints the eigenfrequency table of
// src/MxSolver.cpp:176-196.
//
// Differences from the reference binary: no MPI::Init - one process drives one GPU.  Started N times by any launcher
// that sets RANK / WORLD_SIZE / LOCAL_RANK (torchrun's convention; `mpirun` users export them from their own rank
// variables), the processes share the RCCL unique id through the file named by --id-file (or MXG_ID_FILE): rank 0
// writes it, the others wait for it, and the deck is solved on a decomposed grid, one MxGridDomain box per process.
// 1-D/2-D decks are rejected instead of instantiated; the
// eigenvector files of MxIO::save are written as HDF5 when a libhdf5 can be loaded at run time and as
// `.npy` otherwise; `--eigvals=FILE` additionally writes the frequencies as text.
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "mx/MxProblem.h"

int main(int argc, char* argv[]) {
  std::string inFile, eigFile, idFile = getenv("MXG_ID_FILE") ? getenv("MXG_ID_FILE") : "";
  int device = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0;
  const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0, nranks = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
  bool strict = true, fields = true;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a.compare(0, 9, "--infile=") == 0) inFile = a.substr(9);
    else if (a == "--infile" && i + 1 < argc) inFile = argv[++i];
    else if (a.compare(0, 10, "--eigvals=") == 0) eigFile = a.substr(10);
    else if (a.compare(0, 9, "--device=") == 0) device = atoi(a.substr(9).c_str());
    else if (a.compare(0, 10, "--id-file=") == 0) idFile = a.substr(10);
    else if (a == "--fix-periodic-wrap") strict = false;
    else if (a == "--no-fields") fields = false;
    else if (a == "--help" || a == "-h") {
      std::cout << "Usage: " << argv[0] << " [options]\n  options:\n"
                << "  --help               Prints this help message\n"
                << "  --infile    string   XML format input file.\n"
                << "  --eigvals   string   write 'index real imag error' lines of the eigenfrequencies (Hz)\n"
                << "  --device    int      GPU ordinal (default LOCAL_RANK, else 0)\n"
                << "  --id-file   string   WORLD_SIZE > 1: file through which rank 0 hands the RCCL unique id to the others\n"
                << "  --no-fields          do not write the eigenvector files mxYeeFit{B,E}Field_eigVecs_vecNN.h5 (.npy without HDF5)\n"
                << "  --fix-periodic-wrap  keep the wrap-around entries the reference drops from curlE on\n"
                << "                       periodic axes (DESIGN.md section 7); default: reference behaviour\n";
      return 0;
    } else {
      std::cout << "Unrecognised option: " << a << "\n";
      return -1;
    }
  }
  if (inFile == "") {
    std::cout << "Please specify an input file using --infile=your_file.mx\n";
    return 0;
  }
  try {
    MxUtil::XML::Node xmlObj = MxUtil::XML::Node::parseFile(inFile);
    int dim = atoi(MxUtil::XML::getAttr("dim", xmlObj, "0").c_str());
    if (dim < 1 || dim > 3) {
      std::cout << "Simulation dimension invalid or not given, using 3D.\n";
      dim = 3;
    }
    std::string domain = MxUtil::XML::getAttr("domain", xmlObj, "");
    if (domain != "frequency" && domain != "time") {
      std::cout << "Simulation domain invalid or not given, using frequency-domain.\n";
      domain = "frequency";
    }
    if (dim != 3) {
      std::cout << "Only 3-D decks run on this build (got dim = " << dim << ").\n";
      return 2;
    }
    RCP<MxComm> myComm;
    if (nranks > 1 || idFile != "") {
      if (idFile == "") { std::cout << "WORLD_SIZE = " << nranks << " needs --id-file=PATH (or MXG_ID_FILE) on a file system all ranks see\n"; return 2; }
      char id[MXG_UNIQUE_ID_BYTES];
      if (rank == 0) {
        if (mxg_get_unique_id(id) != 0) { std::cout << "maxwell: " << mxg_last_error() << "\n"; return 1; }
        const std::string tmp = idFile + ".tmp";
        { std::ofstream f(tmp.c_str(), std::ios::binary); f.write(id, sizeof id); }
        if (std::rename(tmp.c_str(), idFile.c_str()) != 0) { std::cout << "maxwell: cannot write " << idFile << "\n"; return 1; }
      } else {
        bool got = false;
        for (int tries = 0; tries < 1200 && !got; ++tries) {       // up to two minutes
          std::ifstream f(idFile.c_str(), std::ios::binary);
          if (f && f.read(id, sizeof id)) got = true;
          else { struct timespec ts = {0, 100000000}; nanosleep(&ts, nullptr); }
        }
        if (!got) { std::cout << "maxwell: rank " << rank << " found no RCCL id in " << idFile << "\n"; return 1; }
      }
      myComm = rcp(new MxComm(device, rank, nranks, id));
      myComm->barrier();
      if (rank == 0) std::remove(idFile.c_str());
    } else {
      myComm = rcp(new MxComm(device));
    }
    {
      MxProblem<3> prob3d(xmlObj, myComm, strict);
      prob3d.setSaveFields(fields);
      prob3d.solve();
      if (eigFile != "" && rank == 0) {
        std::ofstream out(eigFile.c_str());
        out << std::setprecision(17);
        auto const& f = prob3d.frequencies();
        auto const& e = prob3d.residuals();
        for (size_t i = 0; i < f.size(); ++i) out << i << " " << f[i].real() << " " << f[i].imag() << " " << e[i] << "\n";
      }
    }   // solver, operators and vectors go before the context
  } catch (const std::exception& e) {
    std::cout << e.what() << "\n";
    return 1;
  } catch (int) {
    std::cout << "maxwell: device error: " << mxg_last_error() << "\n";
    return 1;
  }
  return 0;
}
