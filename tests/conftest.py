import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURES = ["toy_bo_d2", "complete_bo_d3", "toy_init_X", "toy_init_Z", "toy_c1_Z50", "causal_d2", "coral_max_d3",
            "graph_ard_d4", "jitter_ladder"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_fixture(name):
    f = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: f[k] for k in f.files}
    d["task"] = str(d["task"])
    d["note"] = str(d["note"])
    for k in ("mX", "vX", "mXs", "vXs"):
        d.setdefault(k, None)
    ls = d["lengthscale"]
    d["lengthscale_arg"] = float(ls[0]) if ls.size == 1 else ls
    return d


@pytest.fixture(params=FIXTURES)
def golden(request):
    d = load_fixture(request.param)
    d["name"] = request.param
    return d


def assert_parity(hip, oracle, truth, what, rtol=1e-5, slack=8.0):
    """Floating-point parity bar of BASELINE.json's north_star: rtol 1e-5 against the fp64 oracle.
    Where Ky is ill-conditioned two correct fp64 implementations differ by eps*sqrt(cond): the 80-bit
    arbiter `truth` measures the oracle's own error and the HIP result may deviate from the oracle by
    at most rtol*|oracle| + slack * (the oracle's worst relative error on this case) * |oracle|."""
    hip, oracle, truth = (np.asarray(a, dtype=np.float64).reshape(-1) for a in (hip, oracle, truth))
    scale = np.maximum(np.abs(truth), 1e-300)
    oracle_rel = np.max(np.abs(oracle - truth) / scale)
    hip_rel = np.max(np.abs(hip - truth) / scale)
    bound = rtol + slack * oracle_rel
    diff_rel = np.max(np.abs(hip - oracle) / np.maximum(np.abs(oracle), 1e-300))
    assert hip_rel <= bound, f"{what}: HIP vs truth {hip_rel:.3e} > {bound:.3e} (oracle vs truth {oracle_rel:.3e})"
    assert diff_rel <= bound + oracle_rel, f"{what}: HIP vs oracle {diff_rel:.3e} (oracle vs truth {oracle_rel:.3e})"
    return hip_rel, oracle_rel, diff_rel
