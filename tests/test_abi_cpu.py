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
