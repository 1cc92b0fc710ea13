import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import clrs_amd
from clrs_amd.mw import solvesdp_mw, MwSchurContext
from clrs_amd.problems import cohnelkies_multi
thr = dict(dual_error_threshold=1e-30, primal_error_threshold=1e-30, duality_gap_threshold=1e-15)
full = clrs_amd.flatten(cohnelkies_multi(8, 15, [1.0 + 0.125 * k for k in range(15)]))
for label, kw, env in (("default", {}, {}), ("pipeline off", dict(pipeline=False), {}), ("words off", {}, {"CLRS_MW_STREAM_WORDS": "0"})):
    for k, v in env.items(): os.environ[k] = v
    ctx = MwSchurContext(full, limbs=5, **kw)
    for rep in range(2):
        t0 = time.time()
        r = solvesdp_mw(full, ctx=ctx, **thr)
        dt = time.time() - t0
    print(label, r.status, r.error_code, r.iterations, "%.3f ms/it" % (1e3 * r.time_total / r.iterations), "wall %.2f s" % dt, flush=True)
    h = r.history
    print("  last rows:", np.array2string(h[-3:, [0, 1, 4, 5, 6, 7, 8, 9]], precision=3, max_line_width=200))
    ctx.close()
    for k in env: os.environ.pop(k)
