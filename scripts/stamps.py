"""Diagnostic: where does one workgroup of k_cluster_assemble spend its cycles?  (s_memtime stamps, 100 MHz ticks)
Build the stamped variant on the CPU box first:  python scripts/stamps.py build ; then run on the GPU box."""
import ctypes as C
import os
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clrs_amd
from clrs_amd import _lib
OUT = os.path.join(_lib.CSRC, "_diag", "libclrs_hip_stamps.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    print(_lib.build(extra_flags=["-DCLRS_FUSED_STAMPS"], out=OUT))
    sys.exit(0)
_lib.load(OUT)
from clrs_amd.solver import SchurContext
from tests.util import flat, spd_iterates, chol_blocks_np
name = sys.argv[1] if len(sys.argv) > 1 else "ce_8_15"
f = flat(name)
if len(sys.argv) > 2:
    import bench
    f = bench.replicate_clusters(f, int(sys.argv[2]))
X, Y = spd_iterates(f, seed=1)
Xc = chol_blocks_np(f, X)
ctx = SchurContext(f)
for _ in range(5):
    ctx.compute_S_integrated(Xc, Y)
st = (C.c_uint64 * 64)()
ctx.L.clrs_debug_stamps(ctx.h, st)
st = np.array(st, dtype=np.float64)
t0 = st[0]
names = ["stage", "TY", "GY", "trsm", "GX", "accum", ""] if not ctx.wave_clusters() else ["loads", "TY", "GY+AY", "trsm", "GX+S", "", ""]
print("values are shader cycles x10 (s_memtime ticks at the shader clock); wave-per-block kernel:", bool(ctx.wave_clusters()))
prev = t0
for b in range(6):
    base = 1 + 10 * b
    if st[base] == 0:
        break
    for i in range(7):
        v = st[base + i]
        if v == 0:
            continue
        print(f"block {b} stamp {i} (+{names[i-1] if i else 'sync'}): {10*(v-prev):8.0f} ns   cum {10*(v-t0):8.0f} ns")
        prev = v
print("end loop:", 10 * (st[60] - t0), "ns; after S write:", 10 * (st[61] - t0), "ns")
