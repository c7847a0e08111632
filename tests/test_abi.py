"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/ttdg_mgm.h declares, and the ctypes table binds exactly that set (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "ttdg_mgm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ttdg_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from ttdg_mgm_amd import _lib
    return ctypes.CDLL(_lib.LIB_PATH)


def test_every_declared_symbol_is_exported(lib):
    syms = header_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), s


def test_exported_functions_are_exactly_the_header(lib):
    """-fvisibility=hidden + the header's visibility pragma: the dynamic symbol table's FUNCTIONS (nm type T) are the header's
    entry points and nothing else - no kernel launch stubs, no internal helpers (the remaining data symbols are the kernel handle
    objects the HIP runtime registers)."""
    import subprocess
    from ttdg_mgm_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    funcs = sorted(line.split()[-1] for line in out.splitlines() if len(line.split()) == 3 and line.split()[1] in "Tt")
    assert funcs == header_symbols(), sorted(set(funcs) ^ set(header_symbols()))


def test_ctypes_table_matches_header():
    from ttdg_mgm_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_version_and_error_string(lib):
    lib.ttdg_version.restype = ctypes.c_int
    lib.ttdg_last_error.restype = ctypes.c_char_p
    assert lib.ttdg_version() == 111
    assert isinstance(lib.ttdg_last_error(), bytes)


def test_argument_validation_needs_no_gpu(lib):
    from ttdg_mgm_amd import _lib
    l = _lib.load()
    # null pointers are rejected before any launch
    rc = l.ttdg_gemm_f32(None, 1, 1, None, 1, 1, None, 1, 1, None, 4, 4, 4, 1.0, 0.0, None)
    assert rc == -1 and b"null" in l.ttdg_last_error()
    with pytest.raises(ValueError):
        _lib.graphs([3, 0, 2])
    with pytest.raises(ValueError):
        _lib.graphs([1] * 65)


def test_product_path_refuses_cpu_tensors():
    import torch
    from ttdg_mgm_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.lap_batched(torch.zeros(1, 3, 3))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ttdg-mgm_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)
