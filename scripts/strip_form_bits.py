"""Same bits from trsm_pair_kernel (256-row pair blocks, the default wherever the padded row count is a multiple of 256) and
trsm_strip8_kernel (CBO_HIP_STRIP_FORM=8): posterior mean / variance / acquisition of a sweep, the winner, and the prediction
gradients (the kernel's SWEEP = false instantiation, forward and reversed factor), over several shapes -- one process per
(form, shape), digests compared.  usage: python scripts/strip_form_bits.py"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %r)
from cbo_with_oop_amd import CausalExpectedImprovement
from cbo_with_oop_amd.GaussianProcessFactory import HipGaussianProcess
from cbo_with_oop_amd.graphs import meshgrid_candidates
n, m, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(n + d)
lo, hi = [-5.0, -5.0, -5.0][:d], [5.0, 20.0, 5.0][:d]
X = rng.uniform(lo, hi, (n, d))
y = np.sin(X).sum(1, keepdims=True) + 0.1 * rng.standard_normal((n, 1))
shape = {1: [m], 2: [m // 64, 64], 3: [m // 256, 16, 16]}[d]
Xs = meshgrid_candidates(list(zip(lo, hi)), shape)
model = HipGaussianProcess(X, y)
res = CausalExpectedImprovement(float(y.min()), "min", model).sweep(Xs, cost=float(d), want_posterior=True, want_acq=True)
dm, dv = model.get_prediction_gradients(Xs[:320])
h = lambda a: hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()[:16]
print(json.dumps({"mean": h(res["mean"]), "var": h(res["var"]), "acq": h(res["acq"]), "best": [res["best_idx"], res["best_val"]],
                  "dmean": h(dm), "dvar": h(dv), "finite": bool(np.isfinite(res["var"]).all())}))
""" % ROOT
shapes = [(200, 1024, 3), (500, 4096, 3), (1000, 2048, 1), (1500, 4096, 2), (4096, 16384, 3)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
bad = 0
for n, m, d in shapes:
    out = {}
    for form in ("8", "2"):
        env = dict(os.environ, CBO_HIP_STRIP_FORM=form)
        r = subprocess.run([sys.executable, "-c", CODE, str(n), str(m), str(d)], env=env, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            print(f"n={n} m={m} d={d} form {form}: FAILED rc={r.returncode}\n{r.stderr[-1500:]}")
            bad += 1
            out[form] = None
            break
        out[form] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    if None in out.values() or len(out) < 2:
        continue
    same = out["8"] == out["2"]
    bad += 0 if same else 1
    print(f"n={n:5d} m={m:6d} d={d}: {'same bits' if same else 'DIFFERENT'}  strip8 {out['8']}" + ("" if same else f"\n    pair {out['2']}"))
sys.exit(1 if bad else 0)
