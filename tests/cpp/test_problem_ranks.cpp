// The reference's main() on a decomposed grid: `MxProblem<3> prob(xml, comm); prob.solve();` on R ranks, each
// rank a thread of this process with its own context of an in-process world (mxg_local_world) - what
// `maxwell --infile=deck.mx` does per process under a launcher.  Prints the k^2 of rank 0 and checks that every
// rank holds the same eigenvalues.   usage: test_problem_ranks deck.mx R [save]   (save: write the field files)
#include <cstdio>
#include <thread>

#include "mx/MxProblem.h"

int main(int argc, char** argv) {
  if (argc < 3) { std::printf("usage: %s deck.mx nranks\n", argv[0]); return 2; }
  const int R = atoi(argv[2]);
  MxUtil::XML::Node xml = MxUtil::XML::Node::parseFile(argv[1]);
  mxg_local_world* world = nullptr;
  if (mxg_local_world_create(R, &world) != 0) { std::printf("world: %s\n", mxg_last_error()); return 1; }
  std::vector<std::vector<MxComplex> > ev(R);
  std::vector<int> rc(R, 0);
  std::vector<std::thread> pool;
  for (int r = 0; r < R; ++r)
    pool.emplace_back([&, r] {
      mxg_ctx* ctx = nullptr;
      if (mxg_ctx_create_local(0, r, R, world, &ctx) != 0) { rc[r] = 1; return; }
      try {
        RCP<MxComm> comm = rcp(new MxComm(ctx));
        MxProblem<3> prob(xml, comm, true);
        prob.setSaveFields(argc > 3);
        prob.solve();
        ev[r] = prob.eigenvalues();
      } catch (const std::exception& e) {
        std::printf("rank %d: %s\n", r, e.what());
        rc[r] = 1;
      } catch (int) {
        std::printf("rank %d: %s\n", r, mxg_last_error());
        rc[r] = 1;
      }
      mxg_ctx_destroy(ctx);
    });
  for (auto& t : pool) t.join();
  mxg_local_world_destroy(world);
  for (int r = 0; r < R; ++r) if (rc[r]) return 1;
  for (int r = 1; r < R; ++r)
    if (ev[r] != ev[0]) { std::printf("rank %d holds different eigenvalues\n", r); return 1; }
  for (size_t i = 0; i < ev[0].size(); ++i) std::printf("THETA %zu %.17g %.17g\n", i, ev[0][i].real(), ev[0][i].imag());
  return 0;
}
