"""Golden results of the 256-bit CPU oracle (oracle/, the restatement of the reference's algorithm with substitution solves) for the BASELINE
configurations whose oracle runs are too long for the test suite: whole solves with the reference's default options at prec = 256.

    python tests/golden/make_golden_configs.py [names...]      ->  tests/golden/configs_256.npz   (merged into the existing file)

Per instance: iterations, error_code, primal / dual objective, gap, dual / primal error, the history table (HIST_NAMES of oracle/oracle.py)
and the wall time of the run with its thread count.  tests/test_mw_parity.py compares the HIP path at 5 limbs against these numbers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.util import flat
from oracle.oracle import Oracle

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs_256.npz")
CASES = {"ns_8_15_3": {}, "threepoint_3_8_8": dict(omega_p=1e3, omega_d=1e3), "sdpa_x64": {}, "ns_8_15_2": {}, "ce_8_15": {}, "min_f_2": {}}


def main():
    names = sys.argv[1:] or list(CASES)
    old = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for name in names:
        if name == "min_f_2":
            from tests.util import load_flat
            f, _ = load_flat(os.path.join(os.path.dirname(os.path.abspath(__file__)), "min_f_2.npz"))
        else:
            f = flat(name)
        o = Oracle(f, mp_bits=256)
        threads = int(os.environ.get("ORACLE_THREADS", "8"))
        o.set_num_threads(threads)
        t0 = time.time()
        r = o.solvesdp(**CASES[name])
        dt = time.time() - t0
        print("%-18s %3d iterations, code %d, p_obj %.15g d_obj %.15g gap %.3e errors %.3e %.3e, %.1f s on %d threads" % (
            name, r["iterations"], r["error_code"], r["p_obj"], r["d_obj"], r["gap"], r["dual_error"], r["primal_error"], dt, threads), flush=True)
        old[name + "/summary"] = np.array([r["iterations"], r["error_code"], r["p_obj"], r["d_obj"], r["gap"], r["dual_error"], r["primal_error"], dt, threads], dtype=np.float64)
        old[name + "/hist"] = np.asarray(r["hist"], dtype=np.float64)
        np.savez_compressed(OUT, **old)


if __name__ == "__main__":
    main()
