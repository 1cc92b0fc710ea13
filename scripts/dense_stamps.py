"""phase stamps of k_mwx_dense (wave 0 of the first workgroup) on the SDPA x64 instance
(diagnostic build: `CLRS_MW_STAMPS=1 python -c "from clrs_amd import _lib; _lib.build()"` here, then run with
CLRS_HIP_LIB=clusteredlowranksolver.jl_amd/csrc/_diag/libclrs_hip_mwstamps.so; the product library carries no stamps)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.util import flat
from clrs_amd.mw import MwSchurContext
f = flat("sdpa_x64"); K = 5
ctx = MwSchurContext(f, limbs=K)
rng = np.random.default_rng(1)
X, Y = np.zeros((K, f.xy_len)), np.zeros((K, f.xy_len))
for b in range(f.n_blocks):
    n = int(f.block_n[b]); o = int(f.block_off[b])
    for M in (X, Y):
        G = rng.standard_normal((n, n)); M[0, o:o + n * n] = (np.eye(n) + G @ G.T / n).reshape(-1)
tX, tY = torch.tensor(X, device="cuda:0"), torch.tensor(Y, device="cuda:0"); tXc = torch.empty_like(tX)
ctx.cholesky_blocks_dev(tX.data_ptr(), tXc.data_ptr())
ctx.assemble_dev(tXc.data_ptr(), tY.data_ptr())
st = (C.c_uint64 * 16)()
ctx.L.clrs_mw_debug_exact_stamps(ctx.h, None)
ctx.assemble_dev(tXc.data_ptr(), tY.data_ptr())
ctx.L.clrs_mw_debug_exact_stamps(ctx.h, st)
v = [int(x) for x in st]
print("k_mwx_dense stamps (us):", ["%.1f" % ((x - v[0]) / 100.0) for x in v if x])
