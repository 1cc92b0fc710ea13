"""Does the word-based synchronisation between the two streams of the interior-point iteration make progress in a process that has PyTorch's
runtime state (bench.py's situation)?  Prints ms per iteration; meant to be run under `timeout`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.set_device(0)
x = torch.zeros(8, device="cuda:0"); torch.cuda.synchronize()
import clrs_amd
from clrs_amd.mw import MwSchurContext, solvesdp_mw
from clrs_amd.problems import cohnelkies
f = clrs_amd.flatten(cohnelkies(8, 15))
ctx = MwSchurContext(f, limbs=5)
for i in range(4):
    t = time.time()
    r = solvesdp_mw(f, ctx=ctx)
    print(i, r.status, r.iterations, "%.3f ms per iteration (wall %.3f s)" % (1e3 * r.time_total / r.iterations, time.time() - t), flush=True)
for n in (5, 1, 2, 56, 3):
    t = time.time()
    r = solvesdp_mw(f, ctx=ctx, maxiterations=n)
    print("maxiterations", n, r.iterations, r.error_code, "wall %.3f s" % (time.time() - t), flush=True)
ctx.close()
