#!/usr/bin/env python3
"""Generates tests/golden/min_f_2.npz: the instance `min_f(2)` of the reference's examples/PolyOpt.jl:40-86 as this package's
generator builds it (clusteredlowranksolver.jl_amd/problems/polyopt.py::min_f, from the mathematics: S_3-invariant basis of degree
<= 4, 5 x 6 x 7 Chebyshev grid, approximate Fekete points), flattened to the C-ABI layout with its (hi, lo) data, plus the
iteration history of the 256-bit CPU oracle on it.

Why a fixture: this instance is the one the reference documents a solver LOG for (docs/src/solving.md:38-51: mu, objectives,
errors and step lengths of iterations 1-3 and 55-56, the iteration count 56, 77-digit final objectives) -- the tightest vector
the reference holds for the path -- and approximate Fekete selects its 11 sample points with a column-pivoted fp64 QR whose ties
(the grid is symmetric under negation) are decided by rounding: a committed instance keeps the comparison independent of the LAPACK
build.  tests/test_reference_vectors.py checks the generator against this file and the oracle / the HIP path against the log.

Run from the repo root:  python tests/golden/make_golden_reference_vectors.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import clrs_amd  # noqa: E402
from clrs_amd.problems import min_f  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from tests.util import save_flat  # noqa: E402


def main():
    sdp = min_f(2)
    f = clrs_amd.flatten(sdp)
    r = Oracle(f, mp_bits=256).solvesdp()
    assert r["error_code"] == 0
    samples = np.array([[float(t) for t in s] for s in sdp.names["samples"]])
    out = os.path.join(ROOT, "tests", "golden", "min_f_2.npz")
    save_flat(out, f, samples=samples, oracle_hist=r["hist"], oracle_objectives_limbs=r["objectives_limbs"])
    print("wrote", out, "iterations", r["iterations"], "objectives %.17g %.17g" % (r["d_obj"], r["p_obj"]))


if __name__ == "__main__":
    main()
