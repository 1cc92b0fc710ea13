"""phase stamps of k_mwi_step (the workgroup that ends the launch) during a solve of the named problem (diagnostic build:
`CLRS_MW_STAMPS=1 python -c "from clrs_amd import _lib; _lib.build()"`, run with CLRS_HIP_LIB=clusteredlowranksolver.jl_amd/csrc/_diag/libclrs_hip_mwstamps.so)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import flat
from clrs_amd.mw import solvesdp_mw, MwSchurContext
name = sys.argv[1] if len(sys.argv) > 1 else "ce_8_15"
f = flat(name)
from clrs_amd import _lib
for kv in sys.argv[2:]:      # e.g. mw_step_wide=0: library configuration keys set before the context is created
    key, val = kv.split("=")
    _lib.check(_lib.load().clrs_config_set(key.encode(), int(val)))
ctx = MwSchurContext(f, limbs=5)
solvesdp_mw(f, limbs=5, ctx=ctx, maxiterations=5)
st = (C.c_uint64 * 16)()
ctx.L.clrs_mw_debug_exact_stamps(ctx.h, None)
for it in (10, 20, 30):
    solvesdp_mw(f, limbs=5, ctx=ctx, maxiterations=it)
    ctx.L.clrs_mw_debug_exact_stamps(ctx.h, st)
    v = [int(x) for x in st]
    names = ["first wg start", "this wg start", "W panel done", "W in LDS", "Householder done", "Sturm set up", "eig done", "stage 3 done"]
    print(name, "after", it, "iterations:", ", ".join("%s %.2f" % (nm, (x - v[8]) / 100.0) for nm, x in zip(names, v[8:16]) if x))
