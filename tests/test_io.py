"""Field output (MxIO::save layout, src/MxIO.cpp:94-222): HDF5 through a run-time loaded libhdf5, `.npy` without one."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from maxwell_amd import _lib

pytestmark = pytest.mark.usefixtures("built")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(lib, stem, arr):
    dims = (C.c_int64 * 4)(*arr.shape)
    out = C.create_string_buffer(512)
    a = np.ascontiguousarray(arr, dtype=np.float64)
    rc = lib.mxh_field_write(stem.encode(), b"real", dims, a.ctypes.data_as(_lib.c_dp), out, len(out))
    assert rc == 0, lib.mxh_last_error()
    return out.value.decode()


def test_hdf5_round_trip_when_a_library_is_present(tmp_path):
    lib = _lib.load_mxhost()
    if not lib.mxh_hdf5_available():
        pytest.skip("no libhdf5 can be loaded at run time in this environment")
    arr = np.random.default_rng(0).uniform(-1, 1, (5, 4, 3, 3))
    path = _write(lib, str(tmp_path / "mxYeeFitBField_eigVecs_vec00"), arr)
    assert path.endswith(".h5")
    assert open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    dims = (C.c_int64 * 4)()
    back = np.empty(arr.size)
    assert lib.mxh_field_read_h5(path.encode(), b"real", dims, back.ctypes.data_as(_lib.c_dp), back.size) == 0
    assert list(dims) == [5, 4, 3, 3] and np.array_equal(back.reshape(arr.shape), arr)
    assert lib.mxh_field_read_h5(path.encode(), b"imag", dims, None, 0) != 0          # no such dataset
    # a complex field keeps "imag" next to "real" in the same file (src/MxIO.cpp:199-222)
    im = np.random.default_rng(1).uniform(-1, 1, arr.shape)
    out = C.create_string_buffer(512)
    d4 = (C.c_int64 * 4)(*arr.shape)
    assert lib.mxh_field_append(str(tmp_path / "mxYeeFitBField_eigVecs_vec00").encode(), b"imag", d4, im.ctypes.data_as(_lib.c_dp), out, 512) == 0
    assert out.value.decode() == path
    assert lib.mxh_field_read_h5(path.encode(), b"imag", dims, back.ctypes.data_as(_lib.c_dp), back.size) == 0
    assert np.array_equal(back.reshape(arr.shape), im)
    assert lib.mxh_field_read_h5(path.encode(), b"real", dims, back.ctypes.data_as(_lib.c_dp), back.size) == 0
    assert np.array_equal(back.reshape(arr.shape), arr)


def test_npy_fallback_without_hdf5(tmp_path):
    code = (
        "import sys, numpy as np, ctypes as C\n"
        "sys.path.insert(0, %r)\n"
        "from maxwell_amd import _lib\n"
        "lib = _lib.load_mxhost()\n"
        "assert lib.mxh_hdf5_available() == 0\n"
        "arr = np.arange(2 * 3 * 4 * 3, dtype=float).reshape(2, 3, 4, 3)\n"
        "dims = (C.c_int64 * 4)(*arr.shape); out = C.create_string_buffer(512)\n"
        "assert lib.mxh_field_write(%r.encode(), b'real', dims, arr.ctypes.data_as(_lib.c_dp), out, 512) == 0\n"
        "p = out.value.decode(); assert p.endswith('.npy'), p\n"
        "assert np.array_equal(np.load(p), arr)\n"
        "print('NPY_OK')\n" % (ROOT, str(tmp_path / "field")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MX_NO_HDF5="1"))
    assert r.returncode == 0 and "NPY_OK" in r.stdout, r.stdout + r.stderr
