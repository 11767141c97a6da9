"""The C-ABI libraries load without a GPU and export every symbol include/*.h declares."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mx[gh]_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header,libname", [("mxg.h", "libmxg.so"), ("mxh.h", "libmxhost.so")])
def test_exports_every_declared_symbol(built, header, libname):
    if libname == "libmxhost.so":
        C.CDLL(os.path.join(ROOT, "maxwell_amd", "lib", "libmxg.so"), mode=C.RTLD_GLOBAL)
    lib = C.CDLL(os.path.join(ROOT, "maxwell_amd", "lib", libname))
    names = _declared(header)
    assert len(names) > 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_signatures_cover_headers(built):
    from maxwell_amd import _lib
    assert set(_declared("mxg.h")) == set(_lib.MXG_SIGNATURES)
    assert set(_declared("mxh.h")) - {"mxg_ctx"} == set(_lib.MXH_SIGNATURES)


def test_no_cpu_fallback_when_library_missing(monkeypatch, tmp_path):
    """The product path must fail loudly, not route anywhere else."""
    from maxwell_amd import _lib
    monkeypatch.setattr(_lib, "_LIBDIR", str(tmp_path))
    monkeypatch.setattr(_lib, "_cache", {})
    with pytest.raises(_lib.MxError, match="not built"):
        _lib.load_mxg()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "maxwell_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f
