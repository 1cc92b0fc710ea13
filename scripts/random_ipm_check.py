"""The device-resident loop against the 256-bit CPU oracle on random SDPs of awkward shapes (most of them infeasible or unbounded: both must walk
the same trajectory into the same status): the table rows of the first iterations and the status / error code at the end."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clrs_amd
from tests.util import random_simple_sdp
from clrs_amd.mw import solvesdp_mw
from oracle.oracle import Oracle, build
build()
cases = [dict(seed=s, J=3, n_free=s % 4, definite=True) for s in range(4)] + \
        [dict(seed=10, J=2, n_free=5, fixed_P=70, max_n=40, lr_blocks=2), dict(seed=11, J=1, n_free=0, fixed_P=60, max_n=33, lr_blocks=3),
         dict(seed=12, J=40, n_free=2, fixed_P=6, max_n=4, lr_blocks=2), dict(seed=13, J=2, n_free=70, fixed_P=40, max_n=20, lr_blocks=2)]
bad = 0
for kw in cases:
    f = clrs_amd.flatten(random_simple_sdp(**kw))
    r = solvesdp_mw(f, limbs=5, maxiterations=25)
    o = Oracle(f, mp_bits=256); o.set_num_threads(8)
    ro = o.solvesdp(maxiterations=25)
    n = min(len(r.history), len(ro["hist"]))
    rel = 0.0
    for it in range(n):
        for col in (1, 8, 9, 10):                  # mu, alpha_d, alpha_p, beta_c
            a, b = r.history[it, col], ro["hist"][it, col]
            rel = max(rel, abs(a - b) / max(abs(b), 1e-300)) if np.isfinite(a) and np.isfinite(b) else rel
    ok = r.error_code == ro["error_code"] and abs(r.iterations - ro["iterations"]) <= 1 and rel < 1e-6
    bad += not ok
    print(kw, "GPU", r.status, r.error_code, r.iterations, "oracle", ro["error_code"], ro["iterations"], "max rel diff of (mu, alpha, beta) %.1e" % rel, "OK" if ok else "MISMATCH", flush=True)
print("mismatches:", bad)
