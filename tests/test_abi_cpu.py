"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU, exports every
symbol include/clrs_hip.h declares, and the host mirror fails loudly (no fallback) when no device exists."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "clrs_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(clrs_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    import ctypes
    from clrs_amd import _lib
    path = _lib.build()
    L = ctypes.CDLL(path)
    declared = header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/clrs_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared, set(_lib.SYMBOLS) ^ set(declared)


def test_code_object_targets_gfx950_only():
    from clrs_amd import _lib
    blob = open(_lib.build(), "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


def test_error_strings_and_version():
    from clrs_amd import _lib
    L = _lib.load()
    assert b"gfx950" in L.clrs_version()
    assert L.clrs_strerror(0) == b"ok"
    assert b"factorisation" in L.clrs_strerror(3)
    assert b"device" in L.clrs_strerror(-3)


def test_no_cpu_fallback_without_a_device():
    """On a box without a GPU the product path raises; it never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from clrs_amd._lib import ClrsError
    from clrs_amd.solver import SchurContext
    from tests.util import flat
    with pytest.raises(ClrsError, match="no usable HIP device"):
        SchurContext(flat("x2p1"))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "clusteredlowranksolver.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
                assert "libclrs_oracle" not in src and "oracle_create" not in src, fn


def test_flatten_layout_is_the_abi_layout():
    """Column-major blocks, CSR term lists sorted by (p, r, s, rank), transposed partners present (src/solver.jl:1009)."""
    from tests.util import flat
    f = flat("ns_8_3_2")
    assert f.block_off[-1] == int(np.sum(f.block_n.astype(np.int64) ** 2))
    for b in range(f.n_blocks):
        t0, t1 = int(f.term_ptr[b]), int(f.term_ptr[b + 1])
        keys = list(zip(f.term_p[t0:t1], f.term_r[t0:t1], f.term_s[t0:t1], f.term_rank[t0:t1]))
        assert keys == sorted(keys)
        ks = set(keys)
        assert all((p, s, r, k) in ks for (p, r, s, k) in ks)


def test_julia_shim_struct_and_symbols_match_the_header():
    """The Julia shim cannot run here (no Julia): at least its `SdpDesc` must list the fields of `struct clrs_sdp_desc` in the same
    order with matching widths, and every C symbol it binds must be declared in include/clrs_hip.h."""
    hdr = open(os.path.join(ROOT, "include", "clrs_hip.h")).read()
    jl = open(os.path.join(ROOT, "julia", "ClusteredLowRankHIP", "src", "ClusteredLowRankHIP.jl")).read()
    body = re.search(r"typedef struct clrs_sdp_desc \{(.*?)\} clrs_sdp_desc;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    c_fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"(const\s+)?(int32_t|int64_t|double)\s*(\*)?\s*(\w+)$", decl)
        assert m, decl
        c_fields.append((m.group(4), m.group(2), bool(m.group(3))))
    jbody = re.search(r"struct SdpDesc\n(.*?)\nend", jl, flags=re.S).group(1)
    j_fields = []
    for line in jbody.strip().splitlines():
        name, typ = [t.strip() for t in line.split("::")]
        ptr = typ.startswith("Ptr{")
        base = typ[4:-1] if ptr else typ
        j_fields.append((name, {"Int32": "int32_t", "Int64": "int64_t", "Float64": "double"}[base], ptr))
    assert j_fields == c_fields
    declared = set(header_symbols())
    used = set(re.findall(r":(clrs_[a-zA-Z0-9_]+)", jl))
    assert used and used <= declared, used - declared


# ---- every ccall of the Julia package against the C prototypes of include/clrs_hip.h ------------------------------------------------
def _c_prototypes():
    txt = open(os.path.join(ROOT, "include", "clrs_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?\w+\s*\*?)\s*(clrs_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.M):
        args = [a.strip() for a in args.split(",")] if args.strip() not in ("", "void") else []
        protos[name] = (ret.strip(), args)
    return protos


def _julia_type_of(ctype):
    """the Julia ccall type(s) a C parameter type may be bound with"""
    t = re.sub(r"\b\w+\s*(\[\d*\])?$", lambda m: "*" if m.group(1) else "", ctype.strip()) if not ctype.strip().endswith("*") else ctype.strip()
    t = re.sub(r"\s+", " ", t.replace("const ", "")).strip()
    t = t.replace(" *", "*")
    table = {
        "int": {"Cint"}, "int32_t": {"Cint", "Int32"}, "double": {"Cdouble", "Float64"}, "void": {"Cvoid"},
        "double*": {"Ptr{Float64}"}, "int*": {"Ptr{Cint}", "Ref{Cint}"}, "int32_t*": {"Ptr{Int32}", "Ptr{Cint}"}, "void*": {"Ptr{Cvoid}", "Ptr{UInt8}"},
        "char*": {"Cstring"}, "clrs_ctx*": {"Ptr{Cvoid}"}, "clrs_mw_ctx*": {"Ptr{Cvoid}"}, "clrs_ctx**": {"Ref{Ptr{Cvoid}}"}, "clrs_mw_ctx**": {"Ref{Ptr{Cvoid}}"},
        "clrs_sdp_desc*": {"Ref{SdpDesc}"}, "clrs_ipm_data*": {"Ref{IpmData}"}, "clrs_ipm_params*": {"Ref{IpmParams}"}, "clrs_ipm_record*": {"Ref{IpmRecord}", "Ptr{IpmRecord}"}, "clrs_ipm_stop*": {"Ref{IpmStop}"}, "clrs_mw_options*": {"Ref{MwOptions}"}, "clrs_ipm_record_fn": {"Ptr{Cvoid}"},
    }
    return table.get(t)


def test_every_julia_ccall_matches_its_c_prototype():
    """Argument count, argument types and return type of every `ccall` in julia/ClusteredLowRankHIP against the prototype of the symbol in
    include/clrs_hip.h (the Julia package cannot be executed here: this is what keeps it from drifting)."""
    protos = _c_prototypes()
    jl = open(os.path.join(ROOT, "julia", "ClusteredLowRankHIP", "src", "ClusteredLowRankHIP.jl")).read()
    calls = re.findall(r"ccall\(\(([^\n]*?),\s*lib(?:clrs\[\])?\),\s*(\w+),\s*\(([^()]*)\)", jl, flags=re.S)
    assert len(calls) >= 15
    checked = set()
    for symexpr, ret, argt in calls:
        syms = re.findall(r":(clrs_\w+)", symexpr)
        assert syms, symexpr
        jargs = [a.strip() for a in re.split(r",\s*(?![^{]*\})", argt.strip().rstrip(",")) if a.strip()]
        for name in syms:
            assert name in protos, name
            cret, cargs = protos[name]
            assert ret in (_julia_type_of(cret + " x") or _julia_type_of(cret) or set()), (name, ret, cret)
            assert len(jargs) == len(cargs), (name, jargs, cargs)
            for ja, ca in zip(jargs, cargs):
                allowed = _julia_type_of(ca)
                assert allowed is not None, (name, ca)
                assert ja in allowed, (name, ja, ca)
            checked.add(name)
    for must in ("clrs_mw_create_opts", "clrs_mw_schur_assemble", "clrs_mw_schur_factor", "clrs_mw_get_factor", "clrs_mw_schur_solve", "clrs_mw_ipm_create_ex",
                 "clrs_mw_ipm_set_params", "clrs_mw_ipm_init", "clrs_mw_ipm_solve_cb", "clrs_mw_ipm_get", "clrs_mw_ipm_set", "clrs_mw_ipm_objectives"):
        assert must in checked, must


def test_julia_ipm_structs_match_the_header():
    hdr = open(os.path.join(ROOT, "include", "clrs_hip.h")).read()
    jl = open(os.path.join(ROOT, "julia", "ClusteredLowRankHIP", "src", "ClusteredLowRankHIP.jl")).read()
    for cname, jname in (("clrs_ipm_data", "IpmData"), ("clrs_ipm_params", "IpmParams"), ("clrs_ipm_record", "IpmRecord"), ("clrs_ipm_stop", "IpmStop")):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        c_fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"(const\s+)?(int32_t|int64_t|double)\s*(\*)?\s*(.*)$", decl, flags=re.S)
            assert m, decl
            for nm in m.group(4).split(","):
                c_fields.append((nm.strip().lstrip("*").strip(), m.group(2), bool(m.group(3)) or nm.strip().startswith("*")))
        jbody = re.search(r"struct %s\n(.*?)\nend" % jname, jl, flags=re.S).group(1)
        j_fields = []
        for line in jbody.strip().splitlines():
            name, typ = [t.strip() for t in line.split("::")]
            ptr = typ.startswith("Ptr{")
            base = typ[4:-1] if ptr else typ
            j_fields.append((name, {"Int32": "int32_t", "Int64": "int64_t", "Float64": "double"}[base], ptr))
        assert j_fields == c_fields, (cname, j_fields, c_fields)


def test_julia_options_struct_and_matmul_prec():
    """`struct MwOptions` of the Julia package is `clrs_mw_options` field for field (eight int32), and `matmul_prec` is forwarded, not refused."""
    hdr = open(os.path.join(ROOT, "include", "clrs_hip.h")).read()
    jl = open(os.path.join(ROOT, "julia", "ClusteredLowRankHIP", "src", "ClusteredLowRankHIP.jl")).read()
    body = re.sub(r"/\*.*?\*/", "", re.search(r"typedef struct clrs_mw_options \{(.*?)\} clrs_mw_options;", hdr, flags=re.S).group(1), flags=re.S)
    c_names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            m = re.match(r"int32_t\s+(\w+)(\[(\d+)\])?$", decl)
            assert m, decl
            c_names += [m.group(1)] if not m.group(3) else [m.group(1) + str(i + 1) for i in range(int(m.group(3)))]
    jbody = re.search(r"struct MwOptions\n(.*?)\nend", jl, flags=re.S).group(1)
    j_fields = [tuple(t.strip() for t in line.split("::")) for line in jbody.strip().splitlines()]
    assert [n for n, _ in j_fields] == c_names and all(t == "Int32" for _, t in j_fields) and len(c_names) == 8
    from clrs_amd import _lib
    import ctypes
    assert ctypes.sizeof(_lib.MwOptions) == 32 and [n for n, _ in _lib.MwOptions._fields_][:6] == c_names[:6]
    assert "matmul_prec is not available" not in jl and "matmul_limbs=(matmul_prec == prec ? 0 : min(K, limbs_for(matmul_prec)))" in jl


def test_julia_front_end_runs_the_one_call_loop():
    """Round-4 review: the drop-in front end must be the measured path -- `solvesdp` calls clrs_mw_ipm_solve_cb (iterations enqueued one ahead, termination
    on the device) once, prints the table rows from a callback, and no longer loops over clrs_mw_ipm_iterate."""
    jl = open(os.path.join(ROOT, "julia", "ClusteredLowRankHIP", "src", "ClusteredLowRankHIP.jl")).read()
    body = jl[jl.index("function solvesdp(sdp::CLRS.ClusteredLowRankSDP;"):jl.index("function optimize!(")]
    assert ":clrs_mw_ipm_solve_cb" in body and ":clrs_mw_ipm_iterate" not in body and "while true" not in body
    assert "@cfunction(print_row, Cvoid, (Ptr{IpmRecord}, Ptr{Cvoid}))" in body and "function print_row(recp::Ptr{IpmRecord}, user::Ptr{Cvoid})::Cvoid" in jl


def test_julia_front_end_forwards_warm_starts_and_declares_its_optimizer():
    """Round-3 review: `solvesdp(...; dualsol, primalsol)` must reach clrs_mw_ipm_set instead of raising, the constraint renumbering must be the
    package's own helper, and the MOI extension may only assign to a binding its parent declares (Julia >= 1.11 refuses anything else)."""
    jl = open(os.path.join(ROOT, "julia", "ClusteredLowRankHIP", "src", "ClusteredLowRankHIP.jl")).read()
    ext = open(os.path.join(ROOT, "julia", "ClusteredLowRankHIP", "ext", "ClusteredLowRankHIPMOIExt.jl")).read()
    assert "warm starts are not wired" not in jl and ":clrs_mw_ipm_set," in jl and "warm_start_planes" in jl
    assert "function renumber_kept" in jl and "c_removed" not in jl and "cs_leftover" not in jl
    assert "const OPTIMIZER_TYPE" in jl and "function Optimizer(" in jl
    assert "setglobal!" not in ext and "ClusteredLowRankHIP.OPTIMIZER_TYPE[] = Optimizer" in ext
    assert "k != :save_settings" not in jl                      # optimize! passes save_settings on, solvesdp raises for what it cannot do
    proj = open(os.path.join(ROOT, "julia", "ClusteredLowRankHIP", "Project.toml")).read()
    assert "[compat]" in proj and "julia" in proj.split("[compat]")[1]
