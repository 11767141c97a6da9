// `maxwell --infile=your_file.mx`: the reference's command line (src/maxwell.cpp:41-120) on one
// MI355X.  Reads the deck, builds MxProblem<3>, solves and prints the eigenfrequency table of
// src/MxSolver.cpp:176-196.
//
// Differences from the reference binary: no MPI::Init (one process drives one GPU; multi-GPU runs go
// through bench.py / the Python launcher), 1-D/2-D decks are rejected instead of instantiated; the
// eigenvector files of MxIO::save are written as HDF5 when a libhdf5 can be loaded at run time and as
// `.npy` otherwise; `--eigvals=FILE` additionally writes the frequencies as text.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "mx/MxProblem.h"

int main(int argc, char* argv[]) {
  std::string inFile, eigFile;
  int device = 0;
  bool strict = true, fields = true;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a.compare(0, 9, "--infile=") == 0) inFile = a.substr(9);
    else if (a == "--infile" && i + 1 < argc) inFile = argv[++i];
    else if (a.compare(0, 10, "--eigvals=") == 0) eigFile = a.substr(10);
    else if (a.compare(0, 9, "--device=") == 0) device = atoi(a.substr(9).c_str());
    else if (a == "--fix-periodic-wrap") strict = false;
    else if (a == "--no-fields") fields = false;
    else if (a == "--help" || a == "-h") {
      std::cout << "Usage: " << argv[0] << " [options]\n  options:\n"
                << "  --help               Prints this help message\n"
                << "  --infile    string   XML format input file.\n"
                << "  --eigvals   string   write 'index real imag error' lines of the eigenfrequencies (Hz)\n"
                << "  --device    int      GPU ordinal (default 0)\n"
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
    RCP<MxComm> myComm = rcp(new MxComm(device));
    {
      MxProblem<3> prob3d(xmlObj, myComm, strict);
      prob3d.setSaveFields(fields);
      prob3d.solve();
      if (eigFile != "") {
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
