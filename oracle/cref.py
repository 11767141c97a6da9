"""Oracle (TEST INFRASTRUCTURE ONLY): loader for the plain-C CSR row loop (spmv_ref.c) and
numpy restatements of the multivector block operations the reference forwards to Epetra
(src/MxMultiVector.cpp:86-227, src/MxAnasaziMV.cpp:8-213)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def build():
    """Content-stamped like the product build (snapshot copies do not keep mtimes)."""
    import hashlib
    src = os.path.join(_HERE, "spmv_ref.c")
    so = os.path.join(_HERE, "liboracle.so")
    stamp = so + ".stamp"
    digest = hashlib.sha256(open(src, "rb").read() + open(os.path.join(_HERE, "Makefile"), "rb").read()).hexdigest()
    if os.path.exists(so) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return
    subprocess.run(["make", "-B", "-C", _HERE, "liboracle.so"], check=True, stdout=subprocess.DEVNULL)
    open(stamp, "w").write(digest)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def spmv(m, x, threads=False):
    """y = A x with the serial row loop (bit-level reference for the GPU kernel).
    m: scipy CSR (real or complex), x: (ncols,) or (ncols, k)."""
    m = m.tocsr()
    rowptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
    col = np.ascontiguousarray(m.indices, dtype=np.int32)
    x = np.asarray(x)
    squeeze = x.ndim == 1
    if squeeze:
        x = x[:, None]
    cx = np.iscomplexobj(m.data) or np.iscomplexobj(x)
    dt = np.complex128 if cx else np.float64
    val = np.ascontiguousarray(m.data, dtype=dt)
    y = np.zeros((m.shape[0], x.shape[1]), dtype=dt, order="F")
    L = lib()
    for j in range(x.shape[1]):
        xj = np.ascontiguousarray(x[:, j], dtype=dt)
        yj = np.zeros(m.shape[0], dtype=dt)
        if cx:
            L.oracle_spmv_complex(C.c_int64(m.shape[0]), _p(rowptr), _p(col), _p(val), _p(xj), _p(yj))
        elif threads:
            L.oracle_spmv_omp(C.c_int64(m.shape[0]), _p(rowptr), _p(col), _p(val), _p(xj), _p(yj))
        else:
            L.oracle_spmv(C.c_int64(m.shape[0]), _p(rowptr), _p(col), _p(val), _p(xj), _p(yj))
        y[:, j] = yj
    return y[:, 0] if squeeze else y


def spmv_raw(nrows, rowptr_p, col_p, val_p, x, y, threads=True):
    """Timed entry for bench.py's cpu_baseline: raw pointers, no copies."""
    L = lib()
    fn = L.oracle_spmv_omp if threads else L.oracle_spmv
    fn(C.c_int64(nrows), rowptr_p, col_p, val_p, _p(x), _p(y))


# ---- multivector block operations (numpy; column-major semantics are irrelevant here) ----
def mv_dot(A, B):
    """res[j] = A_j^H B_j  (MxMultiVector::dot documented semantics, src/MxMultiVector.hpp:52-55)."""
    return np.einsum("ij,ij->j", np.conj(A), B)


def mv_norm2(A):
    return np.sqrt(np.einsum("ij,ij->j", np.conj(A), A).real)


def mv_update(a, A, b, Y):
    """this = a*A + b*this (src/MxMultiVector.cpp:212-227)."""
    return a * A + b * Y


def mv_times_mat_add_mv(alpha, A, B, beta, Y):
    """this = alpha*A*B + beta*this (MvTimesMatAddMv, src/MxAnasaziMV.cpp:8-86)."""
    return alpha * (A @ B) + beta * Y


def mv_trans_mv(alpha, A, Y):
    """B = alpha * A^H * this (MvTransMv, src/MxAnasaziMV.cpp:114-197)."""
    return alpha * (np.conj(A).T @ Y)
