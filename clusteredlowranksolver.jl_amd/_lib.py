"""ctypes binding of the C ABI (include/clrs_hip.h) -> libclrs_hip.so built in csrc/.

The library is the product: there is NO CPU fallback.  Importing this module never touches the GPU;
`load()` raises if the shared library is missing, and every call raises `ClrsError` on a negative
return code.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libclrs_hip.so")

p_d = C.POINTER(C.c_double)
p_i32 = C.POINTER(C.c_int32)
p_i64 = C.POINTER(C.c_int64)


class SdpDesc(C.Structure):
    """struct clrs_sdp_desc"""
    _fields_ = [
        ("n_clusters", C.c_int32), ("n_free", C.c_int32), ("cluster_P", p_i32), ("B", p_d),
        ("n_blocks", C.c_int32), ("block_cluster", p_i32), ("block_m", p_i32), ("block_delta", p_i32),
        ("block_kind", p_i32), ("term_ptr", p_i64), ("term_p", p_i32), ("term_r", p_i32), ("term_s", p_i32),
        ("term_rank", p_i32), ("term_lambda", p_d), ("term_vec_ptr", p_i64), ("term_vs", p_d), ("term_ws", p_d),
        ("dense_ptr", p_i64), ("dense_p", p_i32), ("dense_A_ptr", p_i64), ("dense_A", p_d),
    ]


class Dims(C.Structure):
    """struct clrs_dims"""
    _fields_ = [("xy_len", C.c_int64), ("x_len", C.c_int64), ("S_len", C.c_int64), ("n_terms", C.c_int64),
                ("n_free", C.c_int32), ("n_clusters", C.c_int32), ("n_blocks", C.c_int32), ("reserved", C.c_int32)]


class IpmData(C.Structure):
    """struct clrs_ipm_data"""
    _fields_ = [("C", p_d), ("c", p_d), ("b", p_d), ("maximize", C.c_int32), ("reserved", C.c_int32), ("constant", C.c_double)]


class IpmParams(C.Structure):
    """struct clrs_ipm_params"""
    _fields_ = [("beta_infeasible", C.c_double), ("beta_feasible", C.c_double), ("gamma", C.c_double),
                ("dual_error_threshold", C.c_double), ("primal_error_threshold", C.c_double), ("max_complementary_gap", C.c_double),
                ("step_length_threshold", C.c_double), ("safe_step", C.c_int32), ("corrector_only", C.c_int32)]


class IpmRecord(C.Structure):
    """struct clrs_ipm_record"""
    _fields_ = [("iter", C.c_int32), ("pd_feas", C.c_int32), ("error_code", C.c_int32), ("factor_status", C.c_int32),
                ("cholesky_status", C.c_int32), ("refine_bits", C.c_int32),
                ("mu", C.c_double), ("d_obj", C.c_double), ("p_obj", C.c_double), ("gap", C.c_double), ("dual_error", C.c_double),
                ("primal_error", C.c_double), ("alpha_d", C.c_double), ("alpha_p", C.c_double), ("beta_c", C.c_double),
                ("max_P", C.c_double), ("max_p", C.c_double), ("max_d", C.c_double)]


class IpmStop(C.Structure):
    """struct clrs_ipm_stop"""
    _fields_ = [("duality_gap_threshold", C.c_double), ("need_dual_feasible", C.c_int32), ("need_primal_feasible", C.c_int32),
                ("max_iterations", C.c_int32), ("reserved", C.c_int32)]


RecordFn = C.CFUNCTYPE(None, C.POINTER(IpmRecord), C.c_void_p)      # clrs_ipm_record_fn


class MwOptions(C.Structure):
    """struct clrs_mw_options"""
    _fields_ = [("exact_products", C.c_int32), ("refine", C.c_int32), ("pipeline", C.c_int32), ("refine_predictor", C.c_int32), ("factor_limbs", C.c_int32),
                ("matmul_limbs", C.c_int32), ("reserved", C.c_int32 * 2)]


class ClrsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"clrs error {code}: {msg}")
        self.code = code


# every symbol include/clrs_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "clrs_ctx_create": (C.c_int, [C.POINTER(SdpDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "clrs_ctx_destroy": (None, [C.c_void_p]),
    "clrs_get_dims": (C.c_int, [C.c_void_p, C.POINTER(Dims)]),
    "clrs_get_unique_counts": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, p_i32, p_i32]),
    "clrs_cholesky_blocks": (C.c_int, [C.c_void_p, p_d, p_d]),
    "clrs_cholesky_blocks_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "clrs_schur_assemble": (C.c_int, [C.c_void_p, p_d, p_d, p_d, p_d]),
    "clrs_schur_factor": (C.c_int, [C.c_void_p]),
    "clrs_get_factor": (C.c_int, [C.c_void_p, p_d, p_d, p_d]),
    "clrs_schur_solve": (C.c_int, [C.c_void_p, p_d, p_d, p_d, p_d]),
    "clrs_schur_assemble_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "clrs_schur_factor_dev": (C.c_int, [C.c_void_p]),
    "clrs_schur_solve_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "clrs_schur_factor_local_dev": (C.c_int, [C.c_void_p]),
    "clrs_q_buffer_dev": (C.c_void_p, [C.c_void_p]),
    "clrs_schur_factor_finish_dev": (C.c_int, [C.c_void_p]),
    "clrs_schur_solve_fwd_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "clrs_u_buffer_dev": (C.c_void_p, [C.c_void_p]),
    "clrs_schur_solve_bwd_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "clrs_S_buffer_dev": (C.c_void_p, [C.c_void_p]),
    "clrs_AY_buffer_dev": (C.c_void_p, [C.c_void_p]),
    "clrs_sync_status": (C.c_int, [C.c_void_p]),
    "clrs_sync_status_cholesky": (C.c_int, [C.c_void_p]),
    "clrs_stream": (C.c_void_p, [C.c_void_p]),
    "clrs_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "clrs_get_timings": (C.c_int, [C.c_void_p, p_d]),
    "clrs_get_counters": (C.c_int, [C.c_void_p, p_d, p_d, p_d, p_d]),
    "clrs_set_graph_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "clrs_ipm_create": (C.c_int, [C.c_void_p, C.POINTER(IpmData)]),
    "clrs_ipm_set_params": (C.c_int, [C.c_void_p, C.POINTER(IpmParams)]),
    "clrs_ipm_init": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "clrs_ipm_iterate": (C.c_int, [C.c_void_p, C.POINTER(IpmRecord)]),
    "clrs_ipm_get": (C.c_int, [C.c_void_p, p_d, p_d, p_d, p_d]),
    "clrs_ipm_debug": (C.c_int, [C.c_void_p, p_d]),
    "clrs_config_set": (C.c_int, [C.c_char_p, C.c_int]),
    "clrs_fused_clusters": (C.c_int, [C.c_void_p]),
    "clrs_wave_clusters": (C.c_int, [C.c_void_p]),
    "clrs_wave2_clusters": (C.c_int, [C.c_void_p]),
    "clrs_wave4_clusters": (C.c_int, [C.c_void_p]),
    "clrs_wave5_clusters": (C.c_int, [C.c_void_p]),
    "clrs_debug_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "clrs_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "clrs_set_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "clrs_get_kernel_times": (C.c_int, [C.c_void_p, C.c_int, p_d, p_i64]),
    "clrs_kernel_name": (C.c_char_p, [C.c_int]),
    "clrs_plan_info": (C.c_int, [C.c_void_p, p_i32, p_i32, p_i32]),
    "clrs_mw_create": (C.c_int, [C.POINTER(SdpDesc), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "clrs_mw_create_ex": (C.c_int, [C.POINTER(SdpDesc), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "clrs_mw_create_opts": (C.c_int, [C.POINTER(SdpDesc), C.c_int, C.c_int, C.c_int, C.POINTER(MwOptions), C.POINTER(C.c_void_p)]),
    "clrs_mw_destroy": (None, [C.c_void_p]),
    "clrs_mw_limbs": (C.c_int, [C.c_void_p]),
    "clrs_mw_get_dims": (C.c_int, [C.c_void_p, C.POINTER(Dims)]),
    "clrs_mw_get_unique_count": (C.c_int, [C.c_void_p, C.c_int32, p_i32]),
    "clrs_mw_cholesky_blocks": (C.c_int, [C.c_void_p, p_d, p_d]),
    "clrs_mw_schur_assemble": (C.c_int, [C.c_void_p, p_d, p_d, p_d, p_d]),
    "clrs_mw_schur_factor": (C.c_int, [C.c_void_p]),
    "clrs_mw_get_S": (C.c_int, [C.c_void_p, p_d, p_d]),
    "clrs_mw_debug_exact_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "clrs_mw_debug_pipe_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "clrs_mw_get_factor": (C.c_int, [C.c_void_p, p_d, p_d, p_d]),
    "clrs_mw_schur_solve": (C.c_int, [C.c_void_p, p_d, p_d, p_d, p_d]),
    "clrs_mw_cholesky_blocks_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "clrs_mw_sync_status_cholesky": (C.c_int, [C.c_void_p]),
    "clrs_mw_set_xchol_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "clrs_mw_schur_assemble_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "clrs_mw_schur_factor_dev": (C.c_int, [C.c_void_p]),
    "clrs_mw_sync_status": (C.c_int, [C.c_void_p]),
    "clrs_mw_schur_solve_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "clrs_comm_unique_id": (C.c_int, [C.c_void_p]),
    "clrs_mw_set_shard": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "clrs_mw_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "clrs_mw_comm_destroy": (C.c_int, [C.c_void_p]),
    "clrs_mw_comm_probe": (C.c_int, [C.c_void_p, C.c_int, p_d, p_i32]),
    "clrs_mw_schur_factor_local_dev": (C.c_int, [C.c_void_p]),
    "clrs_mw_q_gather_dev": (C.c_void_p, [C.c_void_p]),
    "clrs_mw_schur_factor_finish_dev": (C.c_int, [C.c_void_p]),
    "clrs_mw_schur_solve_fwd_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "clrs_mw_u_gather_dev": (C.c_void_p, [C.c_void_p]),
    "clrs_mw_schur_solve_bwd_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "clrs_mw_schur_solve_refine_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "clrs_mw_S_buffer_dev": (C.c_void_p, [C.c_void_p]),
    "clrs_mw_AY_buffer_dev": (C.c_void_p, [C.c_void_p]),
    "clrs_mw_stream": (C.c_void_p, [C.c_void_p]),
    "clrs_mw_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "clrs_mw_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "clrs_mw_get_timings": (C.c_int, [C.c_void_p, p_d]),
    "clrs_mw_get_counters": (C.c_int, [C.c_void_p, p_d, p_d, p_d]),
    "clrs_mw_ipm_create": (C.c_int, [C.c_void_p, C.POINTER(IpmData)]),
    "clrs_mw_ipm_create_ex": (C.c_int, [C.c_void_p, C.POINTER(IpmData), C.c_int]),
    "clrs_mw_ipm_set_params": (C.c_int, [C.c_void_p, C.POINTER(IpmParams)]),
    "clrs_mw_ipm_init": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "clrs_mw_ipm_set": (C.c_int, [C.c_void_p, p_d, p_d, p_d, p_d]),
    "clrs_mw_ipm_iterate": (C.c_int, [C.c_void_p, C.POINTER(IpmRecord)]),
    "clrs_mw_ipm_get": (C.c_int, [C.c_void_p, p_d, p_d, p_d, p_d]),
    "clrs_mw_ipm_objectives": (C.c_int, [C.c_void_p, p_d]),
    "clrs_mw_ipm_set_global": (C.c_int, [C.c_void_p, C.c_int, C.c_int, p_i32, p_i32]),
    "clrs_mw_comm_init_side": (C.c_int, [C.c_void_p, C.c_void_p]),
    "clrs_mw_local_group_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "clrs_mw_local_group_destroy": (None, [C.c_void_p]),
    "clrs_mw_comm_init_local": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "clrs_mw_ipm_solve": (C.c_int, [C.c_void_p, C.POINTER(IpmStop), C.POINTER(IpmRecord), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "clrs_mw_ipm_solve_cb": (C.c_int, [C.c_void_p, C.POINTER(IpmStop), RecordFn, C.c_void_p, C.POINTER(IpmRecord), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "clrs_set_last_error": (None, [C.c_char_p]),
    "clrs_strerror": (C.c_char_p, [C.c_int]),
    "clrs_last_error": (C.c_char_p, []),
    "clrs_version": (C.c_char_p, []),
    "clrs_test_gemm": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, p_d, C.c_int, p_d, C.c_int,
                                 C.c_double, p_d, C.c_int]),
    "clrs_test_potrf": (C.c_int, [C.c_int, C.c_int, p_d, C.c_int]),
    "clrs_test_trsm": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, p_d, C.c_int, p_d, C.c_int]),
    "clrs_test_stream": (C.c_int, [C.c_int, C.c_longlong, C.c_longlong, C.c_int, p_d]),
}


def _hipcc(args):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    r = subprocess.run([hipcc, *args], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stderr[-4000:])


# translation units of libclrs_hip.so: (object, source, extra flags, predicate selecting the files it depends on)
_UNITS = (
    ("clrs_hip.o", "clrs_hip.hip", (), lambda f: not f.startswith("clrs_mw")),
    # the multi-word (extended precision) path: error-free transformations must not be contracted into FMAs
    ("clrs_mw.o", "clrs_mw.hip", ("-ffp-contract=off", "-DMW_SPLIT_UNITS"), lambda f: f.startswith("clrs_mw")),
    # the device code of the larger limb counts, one unit each (explicit instantiations: clrs_mw_inst.h), compiled side by side
    *((f"clrs_mw_k{k}.o", "clrs_mw_inst.hip", ("-ffp-contract=off", f"-DMW_INST_K={k}"),
       lambda f: f.startswith("clrs_mw") and f not in ("clrs_mw.hip", "clrs_mw_ipm_host.inc")) for k in (4, 5, 6)),
    *((f"clrs_mw_k{k}p{part}.o", "clrs_mw_inst.hip", ("-ffp-contract=off", f"-DMW_INST_K={k}", f"-DMW_INST_PART={part}"),
       lambda f: f.startswith("clrs_mw") and f not in ("clrs_mw.hip", "clrs_mw_ipm_host.inc")) for k in (8, 10) for part in (1, 2, 3, 4)),
)
_COMMON = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value")


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = None) -> str:
    """Compile csrc/*.hip for gfx950 into csrc/libclrs_hip.so (in-tree, travels with gpurun): one object per translation
    unit (recompiled only when one of its sources changed), then one link.
    `extra_flags` / `out` build a diagnostic variant (e.g. -DCLRS_FUSED_STAMPS) of the fp64 unit beside it."""
    hdr = os.path.join(_HERE, "..", "include", "clrs_hip.h")
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hip.h", ".inc", ".h")))
    objs, jobs = [], []
    # CLRS_MW_STAMPS=1: the diagnostic variant of the multi-word units with phase stamps in k_mwi_Zi / k_mws_pair / k_mwx_dense (compile-time: the
    # product library carries none), built beside the product as csrc/_diag/libclrs_hip_mwstamps.so (objects there too; load it with CLRS_HIP_LIB=...)
    # CLRS_MW_DEFS="-DX ...": an experimental variant of the multi-word units with those definitions, csrc/_diag/libclrs_hip_mwexp.so (same mechanism)
    mw_defs = tuple(os.environ.get("CLRS_MW_DEFS", "").split()) if out is None else ()
    mw_stamps = (bool(os.environ.get("CLRS_MW_STAMPS")) or bool(mw_defs)) and out is None
    if mw_stamps:
        os.makedirs(os.path.join(CSRC, "_diag"), exist_ok=True)
        out = os.path.join(CSRC, "_diag", "libclrs_hip_mwexp.so" if mw_defs else "libclrs_hip_mwstamps.so")
    for obj, src, flags, mine in _UNITS:
        if mw_stamps:
            flags = (*flags, *(mw_defs if mw_defs else ("-DCLRS_MW_STAMPS",))) if src != "clrs_hip.hip" else flags
            obj = os.path.join("_diag", obj) if src != "clrs_hip.hip" else obj
        o = os.path.join(CSRC, obj if out is None or src != "clrs_hip.hip" or mw_stamps else os.path.join("_diag", os.path.basename(out) + ".o"))
        os.makedirs(os.path.dirname(o), exist_ok=True)
        deps = [os.path.join(CSRC, f) for f in files if mine(f)] + [hdr]
        diag = out is not None and src == "clrs_hip.hip" and not mw_stamps
        if force or diag or not os.path.exists(o) or any(os.path.getmtime(o) < os.path.getmtime(s) for s in deps):
            jobs.append((src, [*_COMMON, *flags, *(extra_flags if diag else ()), "-c", "-o", o, os.path.join(CSRC, src)]))
        objs.append(o)
    rebuilt = bool(jobs)
    if jobs:                                                # the units compile side by side (the multi-word one takes minutes)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=len(jobs)) as pool:
            for src, _ in zip((j[0] for j in jobs), pool.map(lambda j: _hipcc(j[1]), jobs)):
                if verbose:
                    print("compiled", src)
    target = out if out is not None else LIB_PATH
    os.makedirs(os.path.dirname(target), exist_ok=True)       # (diagnostic variants live in csrc/_diag/: delete the directory when done, it ships with gpurun)
    if rebuilt or not os.path.exists(target) or any(os.path.getmtime(target) < os.path.getmtime(o) for o in objs):
        _hipcc(["--offload-arch=gfx950", "-fPIC", "-shared", "-o", target, *objs])
    return target


_lib = None


def load(path: str = None):
    """Load libclrs_hip.so; fails loudly when it has not been built (no fallback path exists).
    `path` (before the first load) selects a diagnostic build of the same library."""
    global _lib, LIB_PATH
    if _lib is None:
        if path is None:
            path = os.environ.get("CLRS_HIP_LIB") or None      # a diagnostic / experimental build of the same library
        if path is not None:
            LIB_PATH = path
        if not os.path.exists(LIB_PATH):
            raise ClrsError(-100, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                  f"(the HIP extension is the only compute path)")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64, and a process that loads /opt/rocm's copy first
        # (through this library) and torch's afterwards ends up with two runtimes, the second of which sees no GPU.  Importing
        # torch first makes its copy the one the dynamic linker binds this library to as well (import only: no GPU is touched).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)   # AttributeError here = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code: int) -> int:
    """Raise on negative codes; positive codes (factorisation failures) are returned to the caller."""
    if code < 0:
        L = load()
        raise ClrsError(code, f"{L.clrs_strerror(code).decode()}: {L.clrs_last_error().decode()}")
    return code
