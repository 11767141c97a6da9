"""CPU oracle for the maxwell hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / scipy / plain C) of the reference
algorithm for the path SURVEY.md section 8 names: Yee / Dey-Mittra operator
assembly, the MxMagWaveOp operator chain, CSR SpMV/SpMM, the multivector block
operations and the shift-invert Apply.  Every function cites the reference
file:line it follows (paths relative to the upstream bauerca/maxwell tree).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``maxwell_amd/`` imports it and the
product path never falls back to it.

Parity pinning: the reference's numerical kernels are Trilinos (Epetra, AztecOO,
Anasazi; not vendored, not buildable offline) and its own test-suite holds no
numeric golden vectors for this path (test/AnasaziInterface.cpp:61 checks
algebraic invariants only).  The oracle is therefore pinned against
  * the analytic pillbox spectrum (example/pillbox.py:14-16 geometry),
  * the exact discrete spectrum of the periodic vacuum box,
  * the Anasazi MultiVecTraits invariant list,
  * structural identities of the assembled operators,
and the committed fixtures under tests/golden/ are generated from it by
tests/golden/make_golden.py.
"""
