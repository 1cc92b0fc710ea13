"""Durations of k_gemm_f64_t on the pairing-GEMM shapes of the roofline instances (run under rocprofv3 --kernel-trace --stats)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clrs_amd
from clrs_amd import _lib
L = _lib.load()
dp = lambda a: a.ctypes.data_as(_lib.p_d)
rng = np.random.default_rng(0)
for (M, N, K) in [(4097, 4097, 2049), (2049, 2049, 1025), (4096, 4096, 2048), (2048, 2048, 1024)]:
    A = np.asfortranarray(rng.standard_normal((K, M))); B = np.asfortranarray(rng.standard_normal((K, N))); Cm = np.zeros((M, N), order="F")
    for _ in range(2):
        assert L.clrs_test_gemm(0, 1, 0, M, N, K, 1.0, dp(A), K, dp(B), K, 0.0, dp(Cm), M) == 0
    print(M, N, K, "GFLOP", 2.0 * M * N * K / 1e9, flush=True)
