"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports exactly
the symbols include/moma_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest
import torch  # noqa: F401  (its HIP runtime must be the one our library binds to)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "moma_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(moma_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib_path():
    from moma_amd import build
    return build.build(verbose=False)


def test_header_and_binding_agree():
    from moma_amd import _lib
    assert _declared() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.moma_version.restype = ctypes.c_int
    assert lib.moma_version() == 3
    lib.moma_error_string.restype = ctypes.c_char_p
    assert lib.moma_error_string(-2)


def test_binding_loads_and_refuses_cpu_tensors(lib_path):
    import torch
    from moma_amd import _lib, ops
    _lib.load()
    with pytest.raises(_lib.MomaHipError):
        ops.infonce_fused(torch.zeros(2, 8), torch.zeros(2, 8), torch.zeros(4, 8), 0.15)
    with pytest.raises(_lib.MomaHipError):
        ops.mha(torch.zeros(2, 8), torch.zeros(24, 8), torch.zeros(24), torch.zeros(8, 8), torch.zeros(8), 4)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from moma_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MomaHipError):
        _lib.load()


def test_macro_f1_matches_reference_formula():
    """Model selection metric of the epoch loop (reference train_student_moma.py:522-531)."""
    import numpy as np
    from moma_amd.helper.loops_moma import macro_f1
    cm = np.array([[5, 1, 0], [2, 3, 0], [0, 0, 0]])
    f0 = 2 * (5 / 7) * (5 / 6) / ((5 / 7) + (5 / 6))
    f1 = 2 * (3 / 4) * (3 / 5) / ((3 / 4) + (3 / 5))
    assert abs(macro_f1(cm) - (f0 + f1) / 3) < 1e-12          # a class without true positives counts 0
    assert macro_f1(np.eye(4) * 7) == 1.0
