#!/usr/bin/env python3
"""Headline benchmark: MxMagWaveOp SpMV bandwidth + eigenpairs/s on the pillbox cavity.

    python bench.py --gpus N --steps K --warmup W

A step is one application of the shifted vector Laplacian (vecLapl - shift*dmA, the matrix
every inner Krylov iteration of MxMagWaveOp::Apply multiplies with) to one block of vectors
resident in HBM.  N = 1: example/pillbox.py geometry at 256^3 Yee cells (BASELINE.json configs[1]).
N > 1: one GPU per MxGridDomain sub-box, weak scaling (every rank owns a 256^3-cell box); the `eigen` object then
holds the decomposed solve of the N = 1 eigen problem (256^3 over all ranks: strong scaling).

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (HIP-event timed, against
the 8 TB/s HBM3E peak, algorithmic bytes of SURVEY 8d) and `cpu_baseline` (the oracle's plain-C
row loop on the same matrix, on the host cores of this box).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is achievable


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--cells", dest="n", type=int, default=256, help="Yee cells per side (per GPU)")
    p.add_argument("--k", type=int, default=1, help="vectors in the block (SpMM when > 1)")
    p.add_argument("--kernel", type=lambda v: int(v, 0), default=-1, help="CSR kernel: 0 default, 1 int64 row pointers (mxg_csr_set_kernel)")
    p.add_argument("--eig-n", type=int, default=256, help="grid of the eigenpairs/s measurement (0 = skip)")
    p.add_argument("--eig-timeout", type=int, default=420, help="N > 1: seconds the decomposed eigen-solve may take")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline (0 = skip)")
    p.add_argument("--cpu-eig-n", type=int, default=96, help="grid of the CPU eigenpairs/s sample (same solver over the CPU backend of the ABI; 0 = skip)")
    p.add_argument("--no-compressed", action="store_true", help="skip the compressed-stream / tile-product side measurements")
    p.add_argument("--spmm-k", type=int, default=8, help="block of vectors of the tile-product SpMM side measurement (0 = skip)")
    p.add_argument("--dist-selftest", action="store_true",
                   help="run the N > 1 code path (torch.distributed + RCCL ctx + box assembly + halo plan) on 1 rank")
    return p.parse_args()


def cpu_baseline(sim, shift, nrows, alg_bytes, budget):
    """Oracle row loop (port of Epetra's serial CSR loop) on the SAME matrix, host cores of this box."""
    from oracle import cref
    nr, nc, nz, rp, ci, va = sim.op_raw("vecLapl", shift=shift)
    x = np.random.default_rng(1).uniform(-1, 1, nc)
    y = np.zeros(nr)
    rp_v, ci_v, va_v = C.cast(rp, C.c_void_p), C.cast(ci, C.c_void_p), C.cast(va, C.c_void_p)
    threads = cref.lib().oracle_num_threads()
    cref.spmv_raw(nr, rp_v, ci_v, va_v, x, y, threads=True)      # warm-up / first touch
    t0 = time.perf_counter(); cref.spmv_raw(nr, rp_v, ci_v, va_v, x, y, threads=False)
    serial = time.perf_counter() - t0
    reps, tsum = 0, 0.0
    while tsum < max(budget - 2.0 * serial, 1.0) and reps < 200:
        t0 = time.perf_counter(); cref.spmv_raw(nr, rp_v, ci_v, va_v, x, y, threads=True)
        tsum += time.perf_counter() - t0; reps += 1
    par = tsum / reps
    return {"value": round(alg_bytes / par / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": "same %d-row matrix, %d applications of oracle/spmv_ref.c (OpenMP static rows); "
                      "serial row loop: %.3f GB/s" % (nr, reps, alg_bytes / serial / 1e9),
            "serial_value": round(alg_bytes / serial / 1e9, 3)}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (one per GPU, the
    environment torch.distributed.run would give them), before anything in this process touches the GPU; relay rank
    0's JSON line; exit with the worst child status."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    host_threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   OMP_NUM_THREADS=str(max(1, min(32, host_threads // n))))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout (the JSON line) is drained by a thread; the main thread watches the children: a rank that dies
    # (no such device, a failed collective) must not leave the others waiting in a rendezvous for ever
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(rc == 0 for rc in rcs):
            break
        time.sleep(0.5)
    if failed is not None:
        log("rank %d exited with code %s: stopping the other ranks" % (failed, procs[failed].returncode))
        time.sleep(5.0)                       # let the others fail on their own (their messages are more useful)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=10)
    rcs = [p.returncode for p in procs]
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    if any(rcs):
        log("rank exit codes: %s" % rcs)
    sys.exit((max(abs(rc) for rc in rcs if rc) or 1) if any(rcs) else 0)


def cpu_eigen_baseline(mx, ctx, n):
    """eigenpairs/s of the CPU path on a bounded sample with THE SAME SOLVER: this repository's MxSolver stack (block
    Krylov-Schur, MxMagWaveOp::Apply, multigrid-preconditioned CG, same tolerances) over the CPU backend of the C ABI
    (oracle/cpu_backend: CSR row loops on the host cores, OpenMP threads standing in for one MPI rank per core) at n^3
    cells, next to the GPU run of the same n^3 problem.  (The reference's Trilinos path is not buildable here.)"""
    from oracle import cpu_backend
    from maxwell_amd import _env
    cpu = cpu_backend.eigen(n, threads=_env.host_threads() if hasattr(_env, "host_threads") else None)
    gpu = eigen_bench(mx, ctx, n)
    return {"value": cpu["eigenpairs_per_s"], "unit": "eigenpairs/s", "kind": "same solver, %d^3" % n, "cores": cpu["threads"],
            "sample": "pillbox %d^3, nev 5, tol 1e-8, multigrid-CG inner solves (tol 1e-9): the product's MxSolver stack over "
                      "oracle/cpu_backend (CPU implementation of include/mxg.h), %.1f s, %d + %d inner iterations; the GPU "
                      "figures beside it are for the same %d^3 problem (`eigen` is 256^3)" % (n, cpu["eigensolve_seconds"], cpu["vec_lin_iters"], cpu["sca_lin_iters"], n),
            "eigensolve_seconds": cpu["eigensolve_seconds"], "k2": cpu["k2"],
            "gpu_same_size": {"eigenpairs_per_s": gpu["eigenpairs_per_s"], "eigensolve_seconds": gpu["eigensolve_seconds"],
                              "vec_lin_iters": gpu["vec_lin_iters"], "sca_lin_iters": gpu["sca_lin_iters"], "k2": gpu["k2"]},
            "gpu_over_cpu": round(gpu["eigenpairs_per_s"] / cpu["eigenpairs_per_s"], 1)}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.dist_selftest:
        spawn_ranks(args)
    # stdout carries exactly one line, the JSON: native libraries print banners and progress to fd 1 (RCCL's version
    # block, gloo's rendezvous note, the solver's tables), so fd 1 points at stderr until that line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("ERROR: --gpus %d but WORLD_SIZE %d" % (args.gpus, world))
        sys.exit(2)
    if world > 1 or args.dist_selftest:
        # torch bundles its own ROCm runtime libraries with the same SONAMEs as /opt/rocm's: whichever
        # is loaded first serves the whole process.  Import torch BEFORE libmxg so that both run on one
        # consistent set (the usual situation of a torch extension); N = 1 never imports torch.
        import torch  # noqa: F401
    from maxwell_amd import _env
    if world > 1:
        os.environ.pop("OMP_NUM_THREADS", None)   # torchrun pins it to 1; the host assembly wants its share
    _env.set_omp_threads()                     # the box's CPU share / ranks on this node, not all visible threads
    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    import maxwell_amd as mx

    dist = None
    uid = None
    if world > 1 or args.dist_selftest:
        # Control plane only: rendezvous, the RCCL unique id, setup-time object collectives and the
        # final max/sum of the timings go over gloo on the CPU.  The data path (halo exchange,
        # inner-product all-reduce) is RCCL inside libmxg.  torch never touches the GPU here, so
        # the process holds exactly one HIP/RCCL runtime (the ROCm one libmxg links).
        import torch
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:     # selftest without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("gloo")
        box = [mx.Ctx.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    if os.environ.get("MXG_SHARE_GPU"):        # rehearsal of N ranks on a 1-GPU box (RCCL may refuse duplicates)
        local = 0
    ctx = mx.Ctx(local, rank, world, uid)
    nccl_ranks_seen = ctx.comm_ranks()         # ncclCommCount of the communicator the data path uses
    if world > 1 and nccl_ranks_seen != world:
        log("ERROR: RCCL reports %d ranks, launched %d" % (nccl_ranks_seen, world))
        sys.exit(4)

    shift = 0.05 * (2.0 * np.pi) ** 2          # src/mx.py:711-712 default for the unit box
    t0 = time.perf_counter()
    if world == 1 and not args.dist_selftest:
        sim = mx.Sim.pillbox(args.n)
        sim.form_chain_on_device(ctx)          # the operator chain on the GPU (row f2), bit-identical to the host chain
        A = sim.to_device(ctx, "vecLapl", shift=shift)
        workload = "pillbox %d^3 Yee cells, vecLapl - shift*dmA (example/pillbox.py geometry)" % args.n
        parallelism = "1 GPU"
    else:
        from maxwell_amd import distributed as mxd
        sim, A, workload, parallelism = mxd.bench_setup(ctx, args.n, shift)
    setup_s = time.perf_counter() - t0
    log("rank %d setup %.1f s: rows %d nnz %d" % (rank, setup_s, A.nrows, A.nnz))
    if args.kernel >= 0:
        A.set_kernel(args.kernel)

    X = mx.MultiVec(ctx, A.ncols, args.k)
    Y = mx.MultiVec(ctx, A.nrows, args.k)
    X.random(12345)
    for _ in range(args.warmup):
        A.apply(X, Y)
    ctx.barrier()                              # stream sync (+ RCCL all-reduce across ranks)
    if dist: dist.barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        A.apply(X, Y)
    ev_ms = ctx.timer_stop()                   # HIP events on the ctx stream (waits for the stream)
    ctx.barrier()
    if dist: dist.barrier()
    wall = time.perf_counter() - t0
    alg = A.apply_bytes(args.k)
    rows, nnz = A.nrows, A.nnz
    if dist:
        import torch
        t = torch.tensor([wall, ev_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ev_ms = t.tolist()
        s = torch.tensor([alg, rows, nnz], dtype=torch.float64)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        alg, rows, nnz = [int(v) for v in s.tolist()]
    ms_per_step = wall * 1e3 / args.steps
    value = alg * args.steps / wall / 1e9
    kern_ms = ev_ms / args.steps
    achieved = alg / world / (kern_ms * 1e-3) / 1e9   # per GPU, dominant kernel

    out = {
        "metric": "MxMagWaveOp SpMV achieved HBM GB/s (eigenpairs/s in `eigen`)",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "nccl_ranks_seen": nccl_ranks_seen,
        "config": {"workload": workload, "rows": rows, "nnz": nnz, "block_vectors": args.k,
                   "parallelism": parallelism, "index_types": "int64 rowptr, int32 colidx",
                   "setup_seconds": round(setup_s, 2)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": A.kernel_name(),
                     "kernel_ms": round(kern_ms, 5),
                     "algorithmic_bytes_per_launch": alg // world,
                     "note": "achieved = nnz*12 + rows*(8+8k) + cols*8k bytes (SURVEY 8d CSR formula) / HIP-event time per "
                             "launch; the kernel reads 2-byte in-block row offsets built at setup instead of the "
                             "8-byte rowptr, so measured traffic is ~6 B/row below the formula"},
    }
    # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this same command
    # (scripts/gpu_suite.sh); the latest committed summary for this workload is attached
    hbm_summary = None
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_counters.json")))[::-1]:
            h = json.load(open(f))
            if h.get("workload", {}).get("rows") == rows // max(world, 1) and args.kernel < 0 and args.k == 1:
                out["roofline"]["traffic"] = int(h["hbm_bytes_per_launch"]["total"])
                out["roofline"]["traffic_source"] = os.path.relpath(f, ROOT) + " (FETCH_SIZE x2 + WRITE_SIZE)"
                hbm_summary = (h, os.path.relpath(f, ROOT))
                break
    except Exception:
        pass
    # what a pure stream with the product's byte mix reaches on THIS device in THIS run (the devices of the pool differ
    # by several per cent): the matrix stream + x read once, y written once, no gathers
    try:
        if rank == 0:
            per = alg // world
            wbytes = (A.nrows * 8 * args.k) // 16 * 16
            sms = ctx.stream_ceiling_ms(per - wbytes, wbytes, 20)
            out["roofline"]["stream_ceiling"] = {"GBps": round(per / (sms * 1e-3) / 1e9, 1), "ms": round(sms, 5),
                                                 "frac_of_peak": round(per / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                 "kernel_frac_of_ceiling": round(sms / kern_ms, 4),
                                                 "what": "mxg_bench_stream: the same bytes read (16-byte non-temporal loads) and written once, no gathers, same device, same run"}
    except Exception as e:
        out["roofline"]["stream_ceiling"] = {"error": str(e)}
    if rank == 0 and args.cpu_seconds > 0 and world == 1 and not args.dist_selftest:
        out["cpu_baseline"] = cpu_baseline(sim, shift, A.nrows, alg, args.cpu_seconds)
        if args.cpu_eig_n > 0:
            try:
                out["cpu_baseline"]["eigen"] = cpu_eigen_baseline(mx, ctx, args.cpu_eig_n)
            except Exception as e:
                out["cpu_baseline"]["eigen"] = {"error": str(e)}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0 and world == 1 and not args.dist_selftest and args.kernel < 0 and not args.no_compressed:
        # The same product with the matrix re-encoded at setup (mxg_csr_compress; results bit-identical), reported beside
        # the headline, which stays on the plain CSR arrays: "effective" divides the CSR formula's bytes by the time, the
        # kernels themselves move far fewer.  Two forms: the compressed stream (16-bit band codes + 8-bit value ids,
        # k_spmv_cs: any matrix) and the tile patterns the eigen-driver runs on lattice operators (k_spmv_tile: 16 B of
        # descriptor per 126 rows, patterns through scalar loads, irregular rows as ELL groups in the same launch).
        A.apply(X, Y)
        ref = Y.to_numpy()
        for key, pat in (("compressed_stream", "0"), ("tile_product", None)):
            try:
                if pat is None:
                    os.environ.pop("MXG_PAT", None)
                else:
                    os.environ["MXG_PAT"] = pat
                Ac = sim.to_device(ctx, "vecLapl", shift=shift)
                Yc = mx.MultiVec(ctx, Ac.nrows, args.k)
                t0 = time.perf_counter(); nb, nf, nbytes = Ac.compress(); tcomp = time.perf_counter() - t0
                form, ntile, npat, nirr, cov = Ac.product_form()
                for _ in range(args.warmup):
                    Ac.apply(X, Yc)
                ctx.sync(); ctx.timer_start()
                for _ in range(args.steps):
                    Ac.apply(X, Yc)
                cms = ctx.timer_stop() / args.steps
                same = bool(np.array_equal(ref, Yc.to_numpy()))
                if form == 3:
                    # bytes of the form (descriptors, patterns, ELL groups of the irregular rows) + x read once + y written once
                    moved = nbytes + Ac.nrows * 8 * args.k + Ac.ncols * 8 * args.k
                    out[key] = {"kernel": Ac.kernel_name() + " (wavefront-uniform tile patterns, scalar loads; irregular rows as ELL groups in the same launch)",
                                "kernel_ms": round(cms, 5), "speedup_vs_csr": round(kern_ms / cms, 3),
                                "effective_GBps": round(alg / (cms * 1e-3) / 1e9, 1), "bytes_moved_per_launch": int(moved),
                                "hbm_GBps": round(moved / (cms * 1e-3) / 1e9, 1), "hbm_frac_of_peak": round(moved / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "tiles": ntile, "tile_patterns": npat, "irregular_rows": nirr, "row_coverage": round(cov, 4),
                                "form_bytes": nbytes, "setup_seconds": round(tcomp, 2), "bit_identical_to_csr": same}
                    if hbm_summary and "k_spmv_tile" in hbm_summary[0].get("kernels", {}):
                        tk = hbm_summary[0]["kernels"]["k_spmv_tile"]
                        out[key]["traffic"] = int(tk["total"])
                        out[key]["traffic_source"] = hbm_summary[1] + " (FETCH_SIZE x2 + WRITE_SIZE)"
                    try:
                        # what a pure stream with the same read / write mix reaches on this device in this run
                        wb = Ac.nrows * 8 * args.k
                        tms = ctx.stream_ceiling_ms(moved - wb, wb, 20)
                        out[key]["stream_ceiling"] = {"ms": round(tms, 5), "GBps": round(moved / (tms * 1e-3) / 1e9, 1),
                                                      "kernel_frac_of_ceiling": round(tms / cms, 4),
                                                      "what": "mxg_bench_stream on bytes_moved_per_launch: %d read, %d written" % (moved - wb, wb)}
                    except Exception as e:
                        out[key]["stream_ceiling"] = {"error": str(e)}
                    if args.spmm_k > 1 and args.k == 1:
                        # block of vectors through the tile form: the pattern is loaded once for all columns of a pass
                        kk = args.spmm_k
                        Xk = mx.MultiVec(ctx, Ac.ncols, kk); Yk = mx.MultiVec(ctx, Ac.nrows, kk); Xk.random(777)
                        for _ in range(3):
                            Ac.apply(Xk, Yk)
                        ctx.sync(); ctx.timer_start()
                        nrep = max(5, args.steps // 10)
                        for _ in range(nrep):
                            Ac.apply(Xk, Yk)
                        kms = ctx.timer_stop() / nrep
                        algk = Ac.apply_bytes(kk)
                        out[key]["spmm"] = {"block_vectors": kk, "kernel_ms": round(kms, 5),
                                            "effective_GBps_csr_formula": round(algk / (kms * 1e-3) / 1e9, 1),
                                            "effective_frac_of_peak": round(algk / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                            "bytes_moved_per_launch": int(nbytes + (Ac.nrows + Ac.ncols) * 8 * kk),
                                            "hbm_frac_of_peak": round((nbytes + (Ac.nrows + Ac.ncols) * 8 * kk) / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                        Xk.close(); Yk.close()
                else:
                    moved = nbytes + Ac.nrows * (2 + 8 * args.k) + Ac.ncols * 8 * args.k
                    out[key] = {"kernel": "k_spmv_cs<RgCfg<256,2048>,7>", "kernel_ms": round(cms, 5), "speedup_vs_csr": round(kern_ms / cms, 3),
                                "effective_GBps": round(alg / (cms * 1e-3) / 1e9, 1), "bytes_moved_per_launch": int(moved),
                                "hbm_GBps": round(moved / (cms * 1e-3) / 1e9, 1), "hbm_frac_of_peak": round(moved / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "blocks": nb, "fallback_blocks": nf, "packed_matrix_bytes": nbytes, "csr_matrix_bytes": Ac.nnz * 12,
                                "setup_seconds": round(tcomp, 2), "bit_identical_to_csr": same}
                Yc.close(); Ac.close()
            except Exception as e:
                out[key] = {"error": str(e)}
        os.environ.pop("MXG_PAT", None)
    if args.eig_n > 0 and world == 1 and not args.dist_selftest:
        try:
            out["eigen"] = eigen_bench(mx, ctx, args.eig_n)
        except Exception as e:   # the SpMV line must still print
            out["eigen"] = {"error": str(e)}
    elif args.eig_n > 0:
        # Decomposed eigen-solve of the N = 1 eigen problem (strong scaling: eig_n^3 cells over all ranks).  Its collectives
        # are first exercised on real RCCL here, so a watchdog on every rank makes sure the SpMV line still
        # prints if a rank stops answering: rank 0 prints what it has, every rank leaves.
        import threading

        def bail():
            # a rank stopped answering: the SpMV line still goes out (with the error in it), the status says failure
            if rank == 0:
                out["eigen"] = {"error": "decomposed eigen-solve did not finish within %d s" % args.eig_timeout}
                os.write(json_fd, (json.dumps(out) + "\n").encode())
            os._exit(3)

        try:
            sim.close()          # the window operators of the SpMV workload (host memory) are not needed any more
        except Exception:
            pass
        dog = threading.Timer(args.eig_timeout, bail)
        dog.daemon = True
        dog.start()
        try:
            out["eigen"] = eigen_bench_dist(mx, ctx, args.eig_n, dist)
        except Exception as e:
            out["eigen"] = {"error": str(e)}
        dog.cancel()
    import resource
    log("rank %d host peak RSS %.1f GiB" % (rank, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2.0 ** 20))
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    # orderly teardown: device objects, then the RCCL communicator, then the control plane
    for obj in (X, Y, A):
        obj.close()
    if dist:
        dist.barrier()
    ctx.close()
    if dist:
        dist.destroy_process_group()


def eigen_bench(mx, ctx, n):
    """eigenpairs/s = converged pairs / wall time of MxSolver::solve() (src/MxSolver.cpp:102-113)."""
    def rss_gib():
        try:
            for line in open("/proc/self/status"):
                if line.startswith("VmRSS:"):
                    return int(line.split()[1]) / 2.0 ** 20
        except OSError:
            pass
        return float("nan")
    import resource
    peak = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2.0 ** 20
    r0 = rss_gib()
    t0 = time.perf_counter()
    sim = mx.Sim.pillbox(n)
    t1 = time.perf_counter()
    r1 = rss_gib(); p1 = peak()
    params = mx.SolverParams(nev=5, basis=30, tol=1e-8, shift=0.05 * (2.0 * np.pi) ** 2,
                             lin_type="cg", lin_tol=1e-9, prec_type="amg")
    solver = mx.Solver(ctx, sim, params)
    t2 = time.perf_counter()
    r2 = rss_gib(); p2 = peak()
    nconv = solver.solve()
    t3 = time.perf_counter()
    log("eigen %d^3: assembly %.1f s, solver setup (operator chain, upload, stream encoding, multigrid) %.1f s, "
        "MxSolver::solve() %.1f s (eigen-solve + residual / divergence checks + E fields); host RSS %.1f GiB before, "
        "%.1f after assembly (peak so far %.1f), %.1f after solver setup (peak %.1f)" % (n, t1 - t0, t2 - t1, t3 - t2, r0, r1, p1, r2, p2))
    st = solver.stats()
    res = [solver.result(i) for i in range(nconv)]
    return {"workload": "pillbox %d^3, nev 5, tol 1e-8, block Krylov-Schur shift-invert, multigrid-preconditioned CG inner solves (tol 1e-9; V(2,2), Chebyshev smoother)" % n,
            "assembly_seconds": round(t1 - t0, 2), "solver_setup_seconds": round(t2 - t1, 2),
            "solve_call_seconds": round(t3 - t2, 2),
            "eigenpairs_per_s": round(nconv / st["eigensolve_seconds"], 4), "converged": nconv,
            "eigensolve_seconds": round(st["eigensolve_seconds"], 3), "applies": st["applies"],
            "vec_lin_iters": st["vec_iters"], "sca_lin_iters": st["sca_iters"],
            "k2": [round(r["k2"].real, 10) for r in res],
            "max_rel_residual": max(r["residual"] for r in res) if res else None}


def eigen_bench_dist(mx, ctx, n, dist):
    """The N = 1 eigen problem (pillbox n^3) on a decomposed grid - STRONG scaling, unlike the SpMV line: one rank per
    MxGridDomain box (src/MxGrid.cpp:171-193), row-partitioned operators with RCCL ghost import, all-reduced inner
    products, multigrid levels row-partitioned down to 4e5 rows and replicated below."""
    from maxwell_amd import distributed as mxd
    from maxwell_amd.api import PILLBOX_SHAPE
    world, rank = ctx.nranks, ctx.rank
    t0 = time.perf_counter()
    sim = mxd.BoxSim([n] * 3, rank, world, origin=(-0.5,) * 3, size=(1.0,) * 3, pec_shape=PILLBOX_SHAPE)
    t1 = time.perf_counter()
    params = mx.SolverParams(nev=5, basis=30, tol=1e-8, shift=0.05 * (2.0 * np.pi) ** 2,
                             lin_type="cg", lin_tol=1e-9, prec_type="amg")
    solver = mx.Solver(ctx, sim, params)
    t2 = time.perf_counter()
    nconv = solver.solve()
    st = solver.stats()
    res = [solver.result(i) for i in range(nconv)]
    secs, asm, setup = st["eigensolve_seconds"], t1 - t0, t2 - t1
    if dist is not None and world > 1:
        import torch
        t = torch.tensor([secs, asm, setup], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        secs, asm, setup = t.tolist()
    log("rank %d eigen (decomposed): assembly %.1f s, setup %.1f s, solve %.1f s" % (rank, t1 - t0, t2 - t1, st["eigensolve_seconds"]))
    solver.close()
    sim.close()
    pg = sim.proc_grid
    return {"workload": "pillbox %d^3 on %dx%dx%d MxGridDomain boxes (the N = 1 problem: strong scaling), nev 5, tol 1e-8, block "
                        "Krylov-Schur shift-invert, multigrid-preconditioned CG inner solves (tol 1e-9), levels row-partitioned "
                        "above 4e5 rows" % (n, pg[0], pg[1], pg[2]),
            "scaling": "strong", "assembly_seconds": round(asm, 2), "solver_setup_seconds": round(setup, 2),
            "eigenpairs_per_s": round(nconv / secs, 4), "converged": nconv,
            "eigensolve_seconds": round(secs, 3), "applies": st["applies"], "vec_lin_iters": st["vec_iters"],
            "sca_lin_iters": st["sca_iters"], "k2": [round(r["k2"].real, 10) for r in res],
            "max_rel_residual": max(r["residual"] for r in res) if res else None}


if __name__ == "__main__":
    main()
