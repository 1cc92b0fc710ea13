import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, clrs_amd
from clrs_amd import _lib
OUT = os.path.join(_lib.CSRC, "_diag", "libclrs_hip_stamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    print(_lib.build(extra_flags=["-DCLRS_FUSED_STAMPS", "-DCLRS_IPM_STAMPS"], out=OUT)); sys.exit(0)
_lib.load(OUT)
from tests.util import flat
from clrs_amd.solver import solvesdp_device, SchurContext
f = flat(sys.argv[1] if len(sys.argv) > 1 else "polyopt40")
ctx = SchurContext(f)
solvesdp_device(f, ctx=ctx, maxiterations=5)
out = np.zeros(32)
ctx.L.clrs_ipm_debug(ctx.h, out.ctypes.data_as(_lib.p_d))
names = ["load", "potrf", "2 trsm", "symmetrise", "householder", "sturm"]
for i in range(6):
    print("%-12s %8.0f cycles" % (names[i], out[i + 1] - out[i]))
