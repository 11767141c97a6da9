"""Build the native libraries in-tree (maxwell_amd/lib/*.so).

  libmxg.so     HIP kernels + the C ABI of include/mxg.h       (hipcc --offload-arch=gfx950)
  libmxhost.so  host C++: assembly, drop-in classes, eigen-driver, C ABI of include/mxh.h (g++)
  maxwell       the reference's command line: `maxwell --infile=deck.mx`                (g++)

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the CPU-only
container; the built .so files travel to the GPU box with the repository snapshot.
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "maxwell_amd", "csrc")
LIBDIR = os.path.join(ROOT, "maxwell_amd", "lib")
INCLUDE = os.path.join(ROOT, "include")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

DEVICE_SOURCES = ["mxg_ctx.hip", "mxg_comm.hip", "mxg_mv.hip", "mxg_csr.hip", "mxg_tile.hip", "mxg_spalg.hip", "mxg_solve.hip"]
HOST_SOURCES = ["mx_geometry.cpp", "mx_yee.cpp", "mxh_sim.cpp", "mxh_mg.cpp", "mxh_solver.cpp",
                "mxh_problem.cpp", "mxh_io.cpp"]
CLI_SOURCE = "maxwell.cpp"          # the reference's `maxwell --infile=deck.mx` entry point


def _digest(sources, extra=""):
    h = hashlib.sha256(extra.encode())
    for s in sorted(sources):
        if os.path.isfile(s):
            h.update(os.path.basename(s).encode())
            h.update(open(s, "rb").read())
    return h.hexdigest()


def _stale(target, sources, extra=""):
    """Content hash, not mtime: the snapshot copied to the GPU box does not keep timestamps."""
    stamp = target + ".stamp"
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    return open(stamp).read().strip() != _digest(sources, extra)


def _write_stamp(target, sources, extra=""):
    open(target + ".stamp", "w").write(_digest(sources, extra))


def _deps(directory, extra=()):
    out = [os.path.join(directory, f) for f in os.listdir(directory)]
    out += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    mxdir = os.path.join(INCLUDE, "mx")
    if os.path.isdir(mxdir):
        out += [os.path.join(mxdir, f) for f in os.listdir(mxdir)]
    return out + list(extra)


def _run(cmd, cwd):
    # everything goes to stderr: bench.py must keep stdout for its single JSON line
    print("+", " ".join(cmd), file=sys.stderr, flush=True)
    subprocess.run(cmd, cwd=cwd, check=True, stdout=sys.stderr)


def build_device(force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    src_dir = os.path.join(CSRC, "device")
    target = os.path.join(LIBDIR, "libmxg.so")
    deps = [d for d in _deps(src_dir) if not d.endswith(".so")]
    if not force and not _stale(target, deps):
        return target
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", "-Wno-unused-result", "-ffp-contract=off",
           "-I" + INCLUDE, "-I" + src_dir] + DEVICE_SOURCES + \
          ["-o", target, "-L" + os.path.join(ROCM, "lib"), "-lrccl",
           "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    _run(cmd, src_dir)
    _write_stamp(target, deps)
    return target


def build_host(force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    src_dir = os.path.join(CSRC, "host")
    target = os.path.join(LIBDIR, "libmxhost.so")
    libmxg = os.path.join(LIBDIR, "libmxg.so")
    deps = [d for d in _deps(src_dir) if not d.endswith(".so")]
    dev_stamp = open(libmxg + ".stamp").read() if os.path.exists(libmxg + ".stamp") else ""
    if not force and not _stale(target, deps, dev_stamp) and os.path.exists(os.path.join(LIBDIR, "maxwell")):
        return target
    # -ffp-contract=off: assembled matrix values must not depend on FMA contraction
    cmd = ["g++", "-O2", "-fopenmp", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-I" + INCLUDE, "-I" + os.path.join(INCLUDE, "mx"), "-I" + src_dir] + HOST_SOURCES + \
          ["-o", target, "-L" + LIBDIR, "-lmxg", "-ldl", "-Wl,-rpath,$ORIGIN"]
    _run(cmd, src_dir)
    # the command-line front end, next to the libraries it links
    cli = os.path.join(LIBDIR, "maxwell")
    cmd = ["g++", "-O2", "-fopenmp", "-std=c++17", "-ffp-contract=off",
           "-I" + INCLUDE, "-I" + os.path.join(INCLUDE, "mx"), "-I" + src_dir, CLI_SOURCE,
           "-o", cli, "-L" + LIBDIR, "-lmxhost", "-lmxg", "-Wl,-rpath,$ORIGIN"]
    _run(cmd, src_dir)
    _write_stamp(target, deps, dev_stamp)
    return target


def build_all(force=False):
    build_device(force)
    build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
