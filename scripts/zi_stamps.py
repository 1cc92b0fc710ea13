"""phase stamps of k_mwi_Zi (first workgroup) during a solve of the named problem: products, arrival at the counter, symmetrisation
(diagnostic build: `CLRS_MW_STAMPS=1 python -c "from clrs_amd import _lib; _lib.build()"` here, then run with
CLRS_HIP_LIB=clusteredlowranksolver.jl_amd/csrc/_diag/libclrs_hip_mwstamps.so; the product library carries no stamps)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import flat
from clrs_amd.mw import solvesdp_mw, MwSchurContext
f = flat("ce_8_15")
ctx = MwSchurContext(f, limbs=5)
solvesdp_mw(f, limbs=5, ctx=ctx, maxiterations=5)
st = (C.c_uint64 * 16)()
ctx.L.clrs_mw_debug_exact_stamps(ctx.h, None)
solvesdp_mw(f, limbs=5, ctx=ctx, maxiterations=20)
ctx.L.clrs_mw_debug_exact_stamps(ctx.h, st)
v = [int(x) for x in st]
print("k_mwi_Zi stamps (us from start):", ["%.2f" % ((x - v[0]) / 100.0) for x in v[:8] if x])
