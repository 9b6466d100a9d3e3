"""CPU tests of the C-ABI boundary: the library loads, exports every symbol include/cbo_hip.h declares,
and refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from cbo_with_oop_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cbo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cbo_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/cbo_hip.h but not exported by libcbo_hip.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype in cbo_with_oop_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.cbo_abi_version() == _lib.ABI_VERSION == 5


def test_no_product_import_of_oracle_or_torch_on_the_path():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cbo_with_oop_amd")):
        for fn in files:
            if fn.endswith(".py"):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in re.sub(r"#.*", "", text).replace("no CPU fallback", ""), fn
                if fn != "sharding.py":
                    assert "import torch" not in text, fn


def test_host_side_reductions():
    lib = _lib.load()
    ys = np.array([0.1, 0.7, 0.7, -1.0])
    idx = ctypes.c_int(-1)
    assert lib.cbo_argmax_sets(_lib.dptr(ys), 4, ctypes.byref(idx)) == 0 and idx.value == 1   # first max wins
    vals = np.array([0.5, 0.9, 0.9, 0.2])
    idxs = np.array([10, 700, 300, 5], dtype=np.int64)
    bv, bi = ctypes.c_double(), ctypes.c_int64()
    assert lib.cbo_argmax_pairs(_lib.dptr(vals), idxs.ctypes.data_as(_lib.c_int64_p), 4, ctypes.byref(bv),
                                ctypes.byref(bi)) == 0
    assert (bv.value, bi.value) == (0.9, 300)                                                     # lowest index on ties
    assert lib.cbo_argmax_sets(None, 0, ctypes.byref(idx)) == _lib.CBO_ERR_INVALID
    assert b"bad argument" in lib.cbo_last_error()


def test_fails_loudly_without_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    h = ctypes.c_void_p()
    rc = _lib.load().cbo_init(0, ctypes.byref(h))
    assert rc == _lib.CBO_ERR_NO_DEVICE
    from cbo_with_oop_amd import GaussianProcessFactory, GaussianProcessType
    with pytest.raises(_lib.CboHipError):
        GaussianProcessFactory.create(GaussianProcessType.NON_CAUSAL_GP, np.zeros((3, 1)), np.zeros((3, 1)))


def test_missing_library_is_an_import_error(tmp_path):
    code = ("import os; os.environ['CBO_HIP_LIB']=r'%s'; import cbo_with_oop_amd._lib as l\n"
            "try:\n    l.load()\nexcept ImportError as e:\n    print('IMPORTERROR')\n") % str(tmp_path / "nope.so")
    out = subprocess.run(["python", "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert "IMPORTERROR" in out.stdout, out.stderr


def test_hand_issued_loads_are_not_touched_before_their_wait():
    """trsm_strip8_kernel issues the next block's loads by inline asm (the compiler must not count or wait for them);
    scripts/check_hand_issued_loads.py compiles the file to ISA and verifies that no instruction names a destination
    register between each of those loads and the block end, where the kernel's own stage-top waits have retired it."""
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_hand_issued_loads.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
