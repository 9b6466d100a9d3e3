"""CPU experiment (numpy fp64): does replacing the in-block forward substitution of the strip kernel by a multiplication with
the explicit inverse of the 128x128 (or 64 / 32 / 16) diagonal block of the factor change the accuracy of the posterior?
Emulates the device's block structure; truth = the fixtures' 80-bit arbiter values or oracle/truth.py."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import gp_oracle as O
from oracle import truth as T
import scipy.linalg as sl


def tile_inverse(Lbb, tile):
    """inverse of a lower-triangular block by substitution on the identity, tile x tile diagonal inverses first"""
    n = Lbb.shape[0]
    W = np.zeros_like(Lbb)
    inv = [sl.solve_triangular(Lbb[s:s + tile, s:s + tile], np.eye(min(tile, n - s)), lower=True) for s in range(0, n, tile)]
    R = np.eye(n)
    for k, s in enumerate(range(0, n, tile)):
        e = min(s + tile, n)
        W[s:e] = inv[k] @ R[s:e]
        R[e:] -= Lbb[e:, s:e] @ W[s:e]
    return W


def forward(L, B, blk, inner):
    """V = L^-1 B: blocks of `blk` rows left-looking; inside a block either substitution over `inner` x `inner` tile inverses
    (inner < blk) or one multiplication with the block's explicit inverse (inner == blk, itself built from 16x16 tiles)"""
    n = L.shape[0]
    V = np.zeros_like(B)
    for i0 in range(0, n, blk):
        i1 = min(i0 + blk, n)
        R = B[i0:i1] - L[i0:i1, :i0] @ V[:i0]
        Lbb = L[i0:i1, i0:i1]
        if inner >= blk:
            V[i0:i1] = tile_inverse(Lbb, 16) @ R
        else:
            for s in range(0, i1 - i0, inner):
                e = min(s + inner, i1 - i0)
                W = tile_inverse(Lbb[s:e, s:e], 16)
                V[i0 + s:i0 + e] = W @ R[s:e]
                R[e:] -= Lbb[e:, s:e] @ V[i0 + s:i0 + e]
    return V


def report(name, X, y, Xs, mean_t, var_t, noise=1e-10):
    post = O.fit(X, y)
    L = post.L
    Ks = O.causal_K(post.X, Xs, None, None, post.variance, post.lengthscale, False)
    z = sl.solve_triangular(L, y.reshape(-1), lower=True)
    kss = post.variance
    print(f"== {name}: n={X.shape[0]} m={Xs.shape[0]} cond(L) by diag = {L.diagonal().max() / L.diagonal().min():.2e}")
    Vl = sl.solve_triangular(L, Ks, lower=True)
    rows = [("lapack trtrs", Vl)]
    for inner in (16, 32, 64, 128):
        rows.append((f"block 128, inner {inner}", forward(L, Ks, 128, inner)))
    for lbl, V in rows:
        var = np.clip(kss - (V * V).sum(0), 1e-15, None) + noise
        mu = V.T @ z
        ev = np.max(np.abs(var - var_t.reshape(-1)) / np.abs(var_t.reshape(-1)))
        em = np.max(np.abs(mu - mean_t.reshape(-1))) / np.max(np.abs(y))
        print(f"  {lbl:24s} max rel var err {ev:.3e}   max |mean err| / max|y| {em:.3e}")


if __name__ == "__main__":
    G = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")
    for f in ["toy_c1_Z50", "jitter_ladder", "coral_max_d3", "toy_bo_d2", "complete_bo_d3", "toy_init_Z"]:
        z = np.load(os.path.join(G, f + ".npz"))
        if "mean_truth" not in z.files:
            continue
        report(f, z["X"], z["y"], z["Xs"], z["mean_truth"], z["var_truth"], float(z["noise_var"]))
    # a dense 1-D set of 400 points (ill-conditioned, several row blocks) and a C2-like 3-D set of 1024
    rng = np.random.default_rng(3)
    for name, X, Xs in [("dense 1-D 400", np.sort(rng.uniform(-5, 20, (400, 1)), 0), np.linspace(-5, 20, 300)[:, None]),
                        ("3-D 1024", rng.uniform(-5, 5, (1024, 3)), rng.uniform(-5, 5, (256, 3)))]:
        y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((X.shape[0], 1))
        post = O.fit(X, y)
        mt, vt, _ = T.truth_predict(X, y, Xs, diag_add=1e-10 + 1e-8 + post.jitter)
        report(name, X, y, Xs, mt, vt)
