# ClusteredLowRankHIP.jl -- thin ccall shim binding the reference's hot path to libclrs_hip.so.
#
# NOT a copy of the reference: it depends on ClusteredLowRankSolver.jl, converts its `ClusteredLowRankSDP`
# into the C description once per solve, and provides drop-in methods with the reference's names and
# argument meaning for the two functions on the path:
#
#     compute_T_decomposition!(sdp, S, A_Y, X_inv, Y, ..., cs_map; prec)        (src/solver.jl:1229-1287)
#     the "solve system" stage of compute_search_direction!                     (src/solver.jl:1527-1582)
#
# Host orchestration stays in Julia (Arb).  The device computes in multi-word fp64: a number is `limbs` doubles
# (limbs = 5 covers the reference's default prec = 256; include/clrs_hip.h, clrs_mw_*), an array is PLANAR,
# limb l of element i at [l * len + i].  At the boundary every Arb midpoint is split into its limbs (`limbs_of`),
# results come back as limbs and are summed into the caller's Arb buffers (`arb_of`); the problem data cross the
# boundary as 2 limbs (double-double).  `limbs = 1` selects the plain fp64 entry points (clrs_*), which cannot factor
# the 2d = 30 sphere-packing problems (DESIGN.md section 2).
# This file cannot be executed in the build container (no Julia there); INTEGRATION.md is the contract and
# tests/test_abi_cpu.py checks the struct below against include/clrs_hip.h.
module ClusteredLowRankHIP

using Libdl
using Arblib
import ClusteredLowRankSolver
const CLRS = ClusteredLowRankSolver

const libclrs = Ref{String}(get(ENV, "CLRS_HIP_LIB", "libclrs_hip.so"))

# struct clrs_sdp_desc (include/clrs_hip.h) -- field order and types must match the header
struct SdpDesc
    n_clusters::Int32
    n_free::Int32
    cluster_P::Ptr{Int32}
    B::Ptr{Float64}
    n_blocks::Int32
    block_cluster::Ptr{Int32}
    block_m::Ptr{Int32}
    block_delta::Ptr{Int32}
    block_kind::Ptr{Int32}
    term_ptr::Ptr{Int64}
    term_p::Ptr{Int32}
    term_r::Ptr{Int32}
    term_s::Ptr{Int32}
    term_rank::Ptr{Int32}
    term_lambda::Ptr{Float64}
    term_vec_ptr::Ptr{Int64}
    term_vs::Ptr{Float64}
    term_ws::Ptr{Float64}
    dense_ptr::Ptr{Int64}
    dense_p::Ptr{Int32}
    dense_A_ptr::Ptr{Int64}
    dense_A::Ptr{Float64}
end

# Limb planes of the problem data handed to the library: the reference holds the sampled problem at `prec` bits (convert_to_prec, src/interface.jl:1078-1112),
# so a solve at `limbs` words per number passes its data with `limbs` planes (clrs_mw_create_opts: data_limbs = 1, 2 or limbs); the fp64 entry points take one.
data_limbs_for(limbs::Integer) = limbs == 1 ? 1 : Int(limbs)

"""Device context + the host arrays that back the description (kept alive for the lifetime of the context)."""
mutable struct HipContext
    handle::Ptr{Cvoid}
    limbs::Int                  # 1: fp64 entry points (clrs_*); 2..6, 8, 10: multi-word entry points (clrs_mw_*)
    keep::Vector{Any}
    block_off::Vector{Int}      # offsets of the blocks (j,l) in the xy layout
    block_n::Vector{Int}
    jl::Vector{Tuple{Int,Int}}  # (j,l) of every block in description order
    cluster_P::Vector{Int}
    cluster_off::Vector{Int}
    S_off::Vector{Int}
    term_map::Vector{NTuple{5,Int}}   # (j,l,r,s,idx-in-A_Y[j][l][r,s]) of every term, in term order
    n_free::Int
end

check(code::Integer) = code < 0 ? error("clrs-hip: " * unsafe_string(ccall((:clrs_strerror, libclrs[]), Cstring, (Cint,), code)) *
                                        ": " * unsafe_string(ccall((:clrs_last_error, libclrs[]), Cstring, ()))) : Int(code)

f64(x) = Float64(Arblib.midref(x))

"""Limbs of the midpoint of `x`: successive roundings to Float64 (the subtraction is exact at the working precision)."""
function limbs_of!(out::AbstractMatrix{Float64}, i::Int, x, K::Int)
    r = Arb(x; prec=max(precision(x), 64 * K + 64))
    Arblib.get_mid!(r, r)
    for l in 1:K
        h = Float64(Arblib.midref(r))
        out[i, l] = h                      # column l of a (len x K) Julia matrix = limb plane l of the planar C array
        h == 0 && break
        Arblib.sub!(r, r, Arb(h; prec=precision(r)))
    end
    return out
end

"""Sum of the limbs of element `i` of a planar array as an Arb midpoint at precision `prec`."""
function arb_of(a::AbstractMatrix{Float64}, i::Int, prec::Int)
    r = Arb(0; prec=max(prec, 64 * size(a, 2) + 64))
    for l in size(a, 2):-1:1
        Arblib.add!(r, r, Arb(a[i, l]; prec=precision(r)))
    end
    out = Arb(r; prec)
    Arblib.get_mid!(out, out)
    return out
end

# struct clrs_mw_options (include/clrs_hip.h): a field < 0 = the library's process-wide default; matmul_limbs 0 = the context's limbs
struct MwOptions
    exact_products::Int32
    refine::Int32
    pipeline::Int32
    refine_predictor::Int32
    factor_limbs::Int32
    matmul_limbs::Int32
    reserved1::Int32
    reserved2::Int32
end

"""
    HipContext(sdp, cs_map; device=0, limbs=5, matmul_limbs=0)

Replaces `precompute_matrices_bilinear_pairings` (src/solver.jl:985-1059) and the preallocation block
(src/solver.jl:298-317): flattens `sdp.A[j][l][r,s][p]` (cluster-local constraint indices through `cs_map[j]`,
0-based) and `sdp.B[j]` into `clrs_sdp_desc` and creates the device context.  `limbs = limbs_for(prec)`; `matmul_limbs = limbs_for(matmul_prec)`
(the reference's keyword, src/solver.jl:125: the products that form the pairing matrices in fewer bits; 0 = `limbs`).
"""
function HipContext(sdp::CLRS.ClusteredLowRankSDP, cs_map; device::Integer=0, limbs::Integer=5, matmul_limbs::Integer=0)
    J = length(sdp.A)
    DL = data_limbs_for(limbs)
    cluster_P = Int32[size(sdp.c[j], 1) for j in 1:J]
    N = size(sdp.B[1], 2)
    # data arrays are collected as Arb and split into DL limb planes at the end
    Bv = Any[]
    for j in 1:J, k in 1:N, p in 1:cluster_P[j]
        push!(Bv, sdp.B[j][p, k])
    end
    bc = Int32[]; bm = Int32[]; bd = Int32[]; bk = Int32[]
    term_ptr = Int64[0]; tp = Int32[]; tr = Int32[]; ts = Int32[]; tk = Int32[]; tl = Any[]
    tvp = Int64[0]; tvs = Any[]; tws = Any[]
    dense_ptr = Int64[0]; dp = Int32[]; dAp = Int64[0]; dA = Any[]
    jl = Tuple{Int,Int}[]; block_n = Int[]; term_map = NTuple{5,Int}[]
    for j in 1:J, l in 1:length(sdp.A[j])
        Ajl = sdp.A[j][l]
        m = size(Ajl, 1)
        highrank = any(!(v isa CLRS.LowRankMat) for r in 1:m for s in 1:m for v in values(Ajl[r, s]))
        delta = 0
        push!(jl, (j, l)); push!(bc, j - 1); push!(bk, highrank ? 1 : 0)
        if highrank                                   # src/interface.jl:1001-1007: dense blocks have m == 1
            for p in sort(collect(keys(Ajl[1, 1])))
                M = Ajl[1, 1][p]
                delta = size(M, 1)
                push!(dp, cs_map[j][p] - 1)
                append!(dA, [M[a, b] for b in 1:delta for a in 1:delta])
                push!(dAp, length(dA))
            end
            m = 1
        else
            items = NTuple{4,Int}[]
            for r in 1:m, s in 1:m, p in keys(Ajl[r, s]), k in 1:length(Ajl[r, s][p].lambda)
                push!(items, (cs_map[j][p] - 1, r - 1, s - 1, k - 1))
            end
            sort!(items)
            inv_cs = Dict(v => k for (k, v) in cs_map[j])
            # position of a term's pairing in A_Y[j][l][r,s] (src/solver.jl:1152-1170): A_Y is filled in the order
            # `for p in keys(A[r,s]) for rnk`, so the index is the rank of (p, rnk) in that iteration order
            ay_pos = Dict{NTuple{4,Int},Int}()
            for r in 1:m, s in 1:m
                idx = 0
                for p in keys(Ajl[r, s]), k in 1:length(Ajl[r, s][p].lambda)
                    idx += 1
                    ay_pos[(r, s, cs_map[j][p], k)] = idx
                end
            end
            for (p0, r0, s0, k0) in items
                A = Ajl[r0+1, s0+1][inv_cs[p0+1]]
                delta = length(A.vs[k0+1])
                push!(tp, p0); push!(tr, r0); push!(ts, s0); push!(tk, k0); push!(tl, A.lambda[k0+1])
                append!(tvs, A.vs[k0+1]); append!(tws, A.ws[k0+1])     # Matrix(::LowRankMat) = sum lam * vs * ws' (src/interface.jl:798-800)
                push!(tvp, length(tvs))
                push!(term_map, (j, l, r0 + 1, s0 + 1, ay_pos[(r0 + 1, s0 + 1, p0 + 1, k0 + 1)]))
            end
        end
        push!(bm, m); push!(bd, delta); push!(block_n, m * delta)
        push!(term_ptr, length(tp)); push!(dense_ptr, length(dp))
    end
    planar(v) = (a = zeros(Float64, max(length(v), 1), DL); foreach(i -> limbs_of!(a, i, v[i], DL), eachindex(v)); a)
    Bflat, tlf, tvsf, twsf, dAf = planar(Bv), planar(tl), planar(tvs), planar(tws), planar(dA)
    keep = Any[cluster_P, Bflat, bc, bm, bd, bk, term_ptr, tp, tr, ts, tk, tlf, tvp, tvsf, twsf, dense_ptr, dp, dAp, dAf]
    desc = Ref(SdpDesc(J, N, pointer(cluster_P), pointer(Bflat), length(bc), pointer(bc), pointer(bm), pointer(bd), pointer(bk),
                       pointer(term_ptr), pointer(tp), pointer(tr), pointer(ts), pointer(tk), pointer(tlf), pointer(tvp),
                       pointer(tvsf), pointer(twsf), pointer(dense_ptr), pointer(dp), pointer(dAp), pointer(dAf)))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve keep begin
        if limbs == 1
            check(ccall((:clrs_ctx_create, libclrs[]), Cint, (Ref{SdpDesc}, Cint, Ref{Ptr{Cvoid}}), desc, device, h))
        else
            opts = Ref(MwOptions(-1, -1, -1, -1, -1, Int32(matmul_limbs), 0, 0))
            check(ccall((:clrs_mw_create_opts, libclrs[]), Cint, (Ref{SdpDesc}, Cint, Cint, Cint, Ref{MwOptions}, Ref{Ptr{Cvoid}}), desc, DL, device, limbs, opts, h))
        end
    end
    ctx = HipContext(h[], limbs, keep, cumsum([0; block_n .^ 2]), block_n, jl, Int.(cluster_P), cumsum([0; Int.(cluster_P)]),
                     cumsum([0; Int.(cluster_P) .^ 2]), term_map, N)
    finalizer(c -> ccall((c.limbs == 1 ? :clrs_ctx_destroy : :clrs_mw_destroy, libclrs[]), Cvoid, (Ptr{Cvoid},), c.handle), ctx)
    return ctx
end

"""Smallest limb count whose operations carry `prec` bits (about 53 K - K bits for K limbs)."""
limbs_for(prec::Integer) = prec <= 53 ? 1 : prec <= 104 ? 2 : prec <= 157 ? 3 : prec <= 209 ? 4 : prec <= 262 ? 5 : prec <= 315 ? 6 : prec <= 420 ? 8 : prec <= 525 ? 10 :
                           error("prec = $prec needs more than 10 limbs")

"""Pack a BlockDiagonal of BlockDiagonals of ArbRefMatrix into the planar xy layout (len x limbs, column-major per block)."""
function pack_xy(ctx::HipContext, M)
    out = zeros(Float64, ctx.block_off[end], ctx.limbs)
    for (b, (j, l)) in enumerate(ctx.jl)
        blk = M.blocks[j].blocks[l]
        n = ctx.block_n[b]
        for c in 1:n, r in 1:n
            limbs_of!(out, ctx.block_off[b] + r + (c - 1) * n, blk[r, c], ctx.limbs)
        end
    end
    return out
end

sym(ctx::HipContext, fp64::Symbol, mw::Symbol) = ctx.limbs == 1 ? fp64 : mw

"""
Drop-in for `compute_T_decomposition!` (src/solver.jl:1229-1287).  Same arguments; the Arb scratch arguments
(`bilinear_pairings_*`, `tempX`, `leftvecs`, ..., `part_r`) are accepted and ignored.  Writes `A_Y`, and -- so that a caller
who keeps the reference's own `compute_search_direction!` finds what it reads there (src/solver.jl:1537-1571) -- copies the
factors back into `S` (-> L_j), `LinvB` and `Q` (-> L_Q).  Callers that use `solve_system!` below can pass
`copy_factors=false` and leave the factors on the device.  Throws the reference's `SolverFailure`s.
"""
function compute_T_decomposition!(ctx::HipContext, sdp, S, A_Y, X_inv, Y, bilinear_pairings_Y=nothing, bilinear_pairings_Xinv=nothing,
                                  LinvB=nothing, Q=nothing, args...; prec=precision(S[1]), copy_factors::Bool=true)
    K = ctx.limbs
    Xc, Yf = pack_xy(ctx, X_inv), pack_xy(ctx, Y)
    T = length(ctx.term_map)
    AY = zeros(Float64, max(T, 1), K)
    lib = libclrs[]
    check(ccall((sym(ctx, :clrs_set_timing, :clrs_mw_set_timing), lib), Cint, (Ptr{Cvoid}, Cint), ctx.handle, 1))
    check(ccall((sym(ctx, :clrs_schur_assemble, :clrs_mw_schur_assemble), lib), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), ctx.handle, Xc, Yf, C_NULL, AY))
    for (t, (j, l, r, s, idx)) in enumerate(ctx.term_map)
        A_Y[j][l][r, s][idx, 1] = arb_of(AY, t, prec)
    end
    st = check(ccall((sym(ctx, :clrs_schur_factor, :clrs_mw_schur_factor), lib), Cint, (Ptr{Cvoid},), ctx.handle))
    J = length(S)
    if 0 < st <= J
        throw(CLRS.SolverFailure("S was not decomposed succesfully in block $st, try again with higher precision. If this occurred in the first iteration, remove linear dependencies in the PSD part of the constraints or turn preprocessing on."))
    elseif st == J + 1
        throw(CLRS.SolverFailure("Q was not decomposed correctly. Try restarting with a higher precision. If this occurred in the first iteration, remove linear dependencies between free variables or turn preprocessing on."))
    end
    if copy_factors
        N = ctx.n_free
        Lf = zeros(Float64, ctx.S_off[end], K)
        LBf = zeros(Float64, max(ctx.cluster_off[end] * N, 1), K)
        LQf = zeros(Float64, max(N * N, 1), K)
        check(ccall((sym(ctx, :clrs_get_factor, :clrs_mw_get_factor), lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                    ctx.handle, Lf, N > 0 ? pointer(LBf) : C_NULL, N > 0 ? pointer(LQf) : C_NULL))
        lb_off = 0
        for j in 1:J
            P = ctx.cluster_P[j]
            for q in 1:P, p in 1:P
                S[j][p, q] = arb_of(Lf, ctx.S_off[j] + p + (q - 1) * P, prec)
            end
            if LinvB !== nothing
                for k in 1:N, p in 1:P
                    LinvB[j][p, k] = arb_of(LBf, lb_off + p + (k - 1) * P, prec)
                end
            end
            lb_off += P * N
        end
        if Q !== nothing
            for b in 1:N, a in 1:N
                Q[a, b] = arb_of(LQf, a + (b - 1) * N, prec)
            end
        end
    end
    t = zeros(Float64, 6)
    check(ccall((sym(ctx, :clrs_get_timings, :clrs_mw_get_timings), lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx.handle, t))
    return t[1], t[2], t[3], t[4], t[5]
end

"""
Drop-in for the "solve system" stage of `compute_search_direction!` (src/solver.jl:1527-1582): `dx` (the flat
`ArbRefMatrix(sum P_j, 1)` of src/solver.jl:187,310) and `dy` (`N x 1`) are overwritten; `rhs_x` is the vector
`-d - <A_*, Z>` (`:1518-1523`, length sum P_j), `rhs_y` is `p` -- both Arb column matrices.
"""
function solve_system!(ctx::HipContext, dx, dy, rhs_x, rhs_y; prec=precision(dx))
    K, N, nx = ctx.limbs, ctx.n_free, ctx.cluster_off[end]
    rx = zeros(Float64, nx, K); ry = zeros(Float64, max(N, 1), K)
    for i in 1:nx
        limbs_of!(rx, i, rhs_x[i, 1], K)
    end
    for k in 1:N
        limbs_of!(ry, k, rhs_y[k, 1], K)
    end
    dxf = zeros(Float64, nx, K); dyf = zeros(Float64, max(N, 1), K)
    check(ccall((sym(ctx, :clrs_schur_solve, :clrs_mw_schur_solve), libclrs[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                ctx.handle, rx, ry, dxf, dyf))
    for i in 1:nx
        dx[i, 1] = arb_of(dxf, i, prec)
    end
    for k in 1:N
        dy[k, 1] = arb_of(dyf, k, prec)
    end
    return nothing
end


# =====================================================================================================================
# Drop-in front end: `solvesdp` with the reference's signature and 5-tuple return, no patch to the reference needed.
#
# The reference's `solvesdp(problem; prec, kwargs...)` (src/solver.jl:71-99) builds the ClusteredLowRankSDP, converts it to
# `prec`, and `solvesdp(sdp; ...)` (:100-744) preprocesses, runs the loop and post-processes.  Here the same steps run with the
# reference's own functions for everything outside the loop (ClusteredLowRankSDP, convert_to_prec, preprocess!, postprocess,
# solution_to_bigfloat) and with the device-resident loop of the library (clrs_mw_ipm_*) in place of :348-589.
# =====================================================================================================================

# struct clrs_ipm_data / clrs_ipm_params / clrs_ipm_record (include/clrs_hip.h) -- field order and types must match the header
struct IpmData
    C::Ptr{Float64}
    c::Ptr{Float64}
    b::Ptr{Float64}
    maximize::Int32
    reserved::Int32
    constant::Float64
end
struct IpmParams
    beta_infeasible::Float64
    beta_feasible::Float64
    gamma::Float64
    dual_error_threshold::Float64
    primal_error_threshold::Float64
    max_complementary_gap::Float64
    step_length_threshold::Float64
    safe_step::Int32
    corrector_only::Int32
end
struct IpmRecord
    iter::Int32
    pd_feas::Int32
    error_code::Int32
    factor_status::Int32
    cholesky_status::Int32
    refine_bits::Int32
    mu::Float64
    d_obj::Float64
    p_obj::Float64
    gap::Float64
    dual_error::Float64
    primal_error::Float64
    alpha_d::Float64
    alpha_p::Float64
    beta_c::Float64
    max_P::Float64
    max_p::Float64
    max_d::Float64
end
struct IpmStop
    duality_gap_threshold::Float64
    need_dual_feasible::Int32
    need_primal_feasible::Int32
    max_iterations::Int32
    reserved::Int32
end

# What the per-record callback of clrs_mw_ipm_solve_cb needs between two rows of the iteration table: the reference prints, in the row of an
# iteration, the objectives and the gap of the iterate the iteration STARTED from (src/solver.jl:566-582)
mutable struct TableState
    iter::Int
    t0::Float64
    d_obj::Float64
    p_obj::Float64
    gap::Float64
end
function print_row(recp::Ptr{IpmRecord}, user::Ptr{Cvoid})::Cvoid
    st = unsafe_pointer_to_objref(user)::TableState
    r = unsafe_load(recp)
    CLRS.@printf("%5d %8.1f %11.3e %11.3e %11.3e %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e %10.2e\n", st.iter, time() - st.t0,
                 r.mu, st.d_obj, st.p_obj, st.gap, r.max_P, r.max_p, r.max_d, r.alpha_d, r.alpha_p, r.beta_c)
    if r.error_code == 0
        st.d_obj, st.p_obj, st.gap = r.d_obj, r.p_obj, r.gap
    end
    st.iter += 1
    return nothing
end

"""Cluster-local renumbering of the constraints preprocessing kept: `original row => position among the kept rows` (what `cs_map[j]` is)."""
function renumber_kept(nrows::Integer, removed)
    gone = Set{Int}(removed)
    out = Dict{Int,Int}()
    for p in 1:nrows
        p in gone || (out[p] = length(out) + 1)
    end
    return out
end

"""
Limb planes (len x K) of the warm-start iterate: x and X from `dualsol`, y and Y from `primalsol` (the meaning of src/solver.jl:202-239), in the
layouts of the device loop (x: constraints stacked by cluster; X, Y: blocks in description order, column-major).
"""
function warm_start_planes(sdp, ctx::HipContext, dualsol, primalsol, K::Int)
    nxy, nx, N = ctx.block_off[end], ctx.cluster_off[end], ctx.n_free
    xw = zeros(Float64, max(nx, 1), K); yw = zeros(Float64, max(N, 1), K)
    Xw = zeros(Float64, max(nxy, 1), K); Yw = zeros(Float64, max(nxy, 1), K)
    for ((con, smp), row) in sdp.order_c                             # (constraint of the Problem, sample) -> row of x
        limbs_of!(xw, row, dualsol.x[con][smp], K)
    end
    for (k, name) in enumerate(sdp.free_coeff_names)
        limbs_of!(yw, k, primalsol.freevars[name], K)
    end
    for (b, (j, l)) in enumerate(ctx.jl)
        name = sdp.matrix_coeff_names[j][l]
        m = size(sdp.A[j][l], 1)
        n = ctx.block_n[b]
        dl = div(n, m)
        whole = !(CLRS.Block(name, 1, 1) in sdp.matrix_coeff_blocks)  # a variable without sub-blocks is stored under its bare name
        for r in 1:m, s in 1:r
            key = (whole && r == 1 && s == 1) ? name : CLRS.Block(name, r, s)
            for (sol, W) in ((dualsol, Xw), (primalsol, Yw))
                M = sol.matrixvars[key]
                for cc in 1:dl, rr in 1:dl
                    limbs_of!(W, ctx.block_off[b] + (r - 1) * dl + rr + ((s - 1) * dl + cc - 1) * n, M[rr, cc], K)
                    r == s || limbs_of!(W, ctx.block_off[b] + (s - 1) * dl + cc + ((r - 1) * dl + rr - 1) * n, M[rr, cc], K)   # the mirrored sub-block: the iterate is symmetric
                end
            end
        end
    end
    return xw, yw, Xw, Yw
end

"""
    solvesdp(problem::Problem; prec=precision(BigFloat), device=0, kwargs...)
    solvesdp(sdp::ClusteredLowRankSDP; prec=precision(BigFloat), device=0, kwargs...)

The reference's `solvesdp` (src/solver.jl:42-127: same keywords, same defaults, same return
`status, dualsol, primalsol, solve_time, errorcode`) with the interior-point loop on the GPU at `limbs_for(prec)` words per number.
`dualsol` / `primalsol` warm-start the device loop (`clrs_mw_ipm_set`).  `matmul_prec` selects the limbs of the pairing products (`clrs_mw_options.matmul_limbs`), `correctoronly` is `clrs_ipm_params.corrector_only`.  Keywords without a
device counterpart (`save_settings`, `testing`) are accepted; a non-default value raises an `ArgumentError`, so that a caller never silently gets something else than it asked for.
"""
function solvesdp(problem::CLRS.Problem; prec=precision(BigFloat), kwargs...)
    sdp = CLRS.ClusteredLowRankSDP(problem, prec=prec)          # src/solver.jl:96-98
    sdp = CLRS.convert_to_prec(sdp, prec)
    return solvesdp(sdp; prec=prec, kwargs...)
end

function solvesdp(sdp::CLRS.ClusteredLowRankSDP;
        prec=precision(BigFloat), device::Integer=0,
        maxiterations=500, beta_infeasible=3//10, beta_feasible=1//10, gamma=9//10,
        omega_p=big(10)^10, omega_d=big(10)^10,
        duality_gap_threshold=1e-15, dual_error_threshold=1e-30, primal_error_threshold=1e-30,
        max_complementary_gap=big(10)^100, need_dual_feasible=false, need_primal_feasible=false,
        verbose=true, step_length_threshold=1e-7,
        dualsol::Union{Nothing,CLRS.DualSolution}=nothing, primalsol::Union{Nothing,CLRS.PrimalSolution}=nothing,
        safe_step::Bool=true, correctoronly=false, save_settings=nothing, preprocess=true, matmul_prec=prec, testing=false)
    (save_settings === nothing || (save_settings.iter_interval === nothing && save_settings.time_interval === nothing && save_settings.callback === nothing)) ||
        throw(ArgumentError("save_settings is not available with the HIP backend"))
    matmul_prec <= prec || throw(ArgumentError("matmul_prec must not exceed prec"))
    warm = dualsol !== nothing && primalsol !== nothing               # src/solver.jl:202: only both together are used
    lib = libclrs[]
    K = limbs_for(prec)
    K >= 2 || throw(ArgumentError("the device-resident loop runs at 2 or more limbs (prec > 53)"))
    # preprocessing with the reference's own code (CLRS.preprocess!, src/pre_postprocessing.jl); the constraints it leaves are renumbered per cluster
    rows_before = [size(sdp.B[j], 1) for j in eachindex(sdp.B)]
    cs, var_rels = preprocess ? CLRS.preprocess!(sdp) : ((), nothing)
    cs_map = [renumber_kept(rows_before[j], (t[3] for t in cs if t[2] == j)) for j in eachindex(sdp.B)]
    ctx = HipContext(sdp, cs_map; device=device, limbs=K, matmul_limbs=(matmul_prec == prec ? 0 : min(K, limbs_for(matmul_prec))))
    DL = data_limbs_for(K)
    nxy, nx, N = ctx.block_off[end], ctx.cluster_off[end], ctx.n_free
    # objective data as DL limb planes: C in the xy layout, c in the x layout, b
    Cf = zeros(Float64, max(nxy, 1), DL); cf = zeros(Float64, max(nx, 1), DL); bf = zeros(Float64, max(N, 1), DL)
    for (b, (j, l)) in enumerate(ctx.jl)
        blk = sdp.C.blocks[j].blocks[l]
        n = ctx.block_n[b]
        for cc in 1:n, r in 1:n
            limbs_of!(Cf, ctx.block_off[b] + r + (cc - 1) * n, blk[r, cc], DL)
        end
    end
    for j in eachindex(sdp.c), p in 1:ctx.cluster_P[j]
        limbs_of!(cf, ctx.cluster_off[j] + p, sdp.c[j][p, 1], DL)
    end
    for k in 1:N
        limbs_of!(bf, k, sdp.b[k, 1], DL)
    end
    data = Ref(IpmData(pointer(Cf), pointer(cf), pointer(bf), sdp.maximize ? 1 : 0, 0, f64(sdp.constant)))
    prm = Ref(IpmParams(Float64(beta_infeasible), Float64(beta_feasible), Float64(gamma), Float64(dual_error_threshold),
                        Float64(primal_error_threshold), Float64(max_complementary_gap), Float64(step_length_threshold), safe_step ? 1 : 0, correctoronly ? 1 : 0))
    GC.@preserve Cf cf bf begin
        check(ccall((:clrs_mw_ipm_create_ex, lib), Cint, (Ptr{Cvoid}, Ref{IpmData}, Cint), ctx.handle, data, DL))
    end
    check(ccall((:clrs_mw_ipm_set_params, lib), Cint, (Ptr{Cvoid}, Ref{IpmParams}), ctx.handle, prm))
    check(ccall((:clrs_mw_ipm_init, lib), Cint, (Ptr{Cvoid}, Cdouble, Cdouble), ctx.handle, Float64(omega_p), Float64(omega_d)))
    d_obj0, p_obj0, gap0 = f64(sdp.constant), f64(sdp.constant), 0.0
    if warm
        # the iterate of a previous solve (src/solver.jl:202-239) as limb planes through clrs_mw_ipm_set.  The reference leaves the case of
        # constraints removed by preprocessing open (its TODO at :203: x is indexed by the ORIGINAL constraints); refused here instead of guessed
        isempty(cs) || throw(ArgumentError("warm start of a problem from which preprocessing removed constraints: pass preprocess=false"))
        xw, yw, Xw, Yw = warm_start_planes(sdp, ctx, dualsol, primalsol, K)
        GC.@preserve xw yw Xw Yw begin
            check(ccall((:clrs_mw_ipm_set, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), ctx.handle, xw, N > 0 ? pointer(yw) : Ptr{Float64}(C_NULL), Xw, Yw))
        end
        obj3 = zeros(Float64, 3 * K)                               # objectives of the starting iterate (src/solver.jl:319-321)
        check(ccall((:clrs_mw_ipm_objectives, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx.handle, obj3))
        d_obj0, p_obj0, gap0 = obj3[1], obj3[K + 1], obj3[2 * K + 1]
    end
    if verbose
        CLRS.@printf("%5s %8s %11s %11s %11s %10s %10s %10s %10s %10s %10s %10s\n", "iter", "time(s)", "μ", "D-obj", "P-obj", "gap",
                     "D-error", "d-error", "p-error", "α_d", "α_p", "beta")
    end
    time_start = time()
    iter, error_code = 1, 0
    dual_error = primal_error = Inf
    gap, d_obj, p_obj, pd_feas = gap0, d_obj0, p_obj0, false
    # The whole loop in ONE call (clrs_mw_ipm_solve_cb): the library enqueues iterations one ahead of the record it waits for and the device evaluates the
    # termination test of src/solver.jl:921-950 itself -- the path bench.py measures.  The rows of the iteration table (:566-582) are printed by a callback as
    # the host reads each record; the records come back as well.
    table = TableState(1, time_start, d_obj0, p_obj0, gap0)
    row_cb = @cfunction(print_row, Cvoid, (Ptr{IpmRecord}, Ptr{Cvoid}))
    nrec = max(Int(maxiterations), 1)
    recs = Vector{IpmRecord}(undef, nrec)
    n_it, err = Ref{Cint}(0), Ref{Cint}(0)
    stop = Ref(IpmStop(Float64(duality_gap_threshold), need_dual_feasible ? 1 : 0, need_primal_feasible ? 1 : 0, Int32(maxiterations), 0))
    GC.@preserve table recs begin
        check(ccall((:clrs_mw_ipm_solve_cb, lib), Cint, (Ptr{Cvoid}, Ref{IpmStop}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{IpmRecord}, Cint, Ref{Cint}, Ref{Cint}),
                    ctx.handle, stop, verbose ? row_cb : C_NULL, pointer_from_objref(table), recs, nrec, n_it, err))
    end
    error_code = Int(err[])
    for i in 1:min(Int(n_it[]), nrec)
        r = recs[i]
        dual_error, primal_error, pd_feas = r.dual_error, r.primal_error, r.pd_feas != 0
        if r.error_code != 0                                       # 1 SolverFailure, 3 mu too large, 4 step too short (docs/src/solving.md:64-70); 5: clrs_hip.h
            if verbose && r.error_code == 1
                J = length(sdp.A)
                if r.cholesky_status > 0
                    j, l = ctx.jl[r.cholesky_status]
                    println("The cholesky decomposition of X was not computed correctly in block ($j,$l). Try again with higher precision")
                elseif 0 < r.factor_status <= J
                    println("S was not decomposed succesfully in block $(r.factor_status), try again with higher precision. If this occurred in the first iteration, remove linear dependencies in the PSD part of the constraints or turn preprocessing on.")
                elseif r.factor_status == J + 1
                    println("Q was not decomposed correctly. Try restarting with a higher precision. If this occurred in the first iteration, remove linear dependencies between free variables or turn preprocessing on.")
                end
                println("We return the current solution and optimality status.")
            end
            break
        end
        d_obj, p_obj, gap = r.d_obj, r.p_obj, r.gap
        iter += 1
    end
    if verbose
        if error_code == 2                                         # src/solver.jl:362-366
            println("The maximum number of iterations has been reached.")
        elseif error_code == 0 && !correctoronly && dual_error < dual_error_threshold && primal_error < primal_error_threshold && gap < duality_gap_threshold
            println("Optimal solution found")
        end
    end
    time_total = time() - time_start
    # the iterate back as Arb midpoints in the reference's containers
    xf = zeros(Float64, max(nx, 1), K); yf = zeros(Float64, max(N, 1), K); Xf = zeros(Float64, max(nxy, 1), K); Yf = zeros(Float64, max(nxy, 1), K)
    check(ccall((:clrs_mw_ipm_get, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), ctx.handle, xf, yf, Xf, Yf))
    x = Arblib.ArbRefMatrix(nx, 1; prec=prec); y = Arblib.ArbRefMatrix(N, 1; prec=prec)
    for i in 1:nx
        x[i, 1] = arb_of(xf, i, prec)
    end
    for k in 1:N
        y[k, 1] = arb_of(yf, k, prec)
    end
    unpack(Mf) = CLRS.BlockDiagonal([CLRS.BlockDiagonal([begin
                b = findfirst(==((j, l)), ctx.jl); n = ctx.block_n[b]
                M = Arblib.ArbRefMatrix(n, n; prec=prec)
                for cc in 1:n, r in 1:n
                    M[r, cc] = arb_of(Mf, ctx.block_off[b] + r + (cc - 1) * n, prec)
                end
                M
            end for l in eachindex(sdp.A[j])]) for j in eachindex(sdp.A)])
    X, Y = unpack(Xf), unpack(Yf)
    if preprocess
        x, y = CLRS.postprocess(x, y, cs, var_rels)                # src/solver.jl:630-632
    end
    dsol, psol = CLRS.solution_to_bigfloat(X, x, Y, y, sdp)
    status = if pd_feas && gap < duality_gap_threshold              # src/solver.jl:727-741
        CLRS.Optimal()
    elseif (pd_feas && gap < 1e-8) || (dual_error < 1e-15 && primal_error < 1e-15 && gap < 1e-8)
        CLRS.NearOptimal()
    elseif pd_feas
        CLRS.Feasible()
    elseif primal_error < primal_error_threshold
        CLRS.PrimalFeasible()
    elseif dual_error < dual_error_threshold
        CLRS.DualFeasible()
    else
        CLRS.NotConverged()
    end
    return status, dsol, psol, time_total, error_code
end

"""
    ClusteredLowRankHIP.optimize!(opt)

`MOI.optimize!` for the reference's `ClusteredLowRankSolver.Optimizer` (ext/MOIExt.jl:395-407) with this package's `solvesdp`: reads
`opt.problem` and `opt.options`, fills `opt.result_data` under the keys the reference's result getters read (`:primalsol`, `:dualsol`,
`:status`, `:errorcode`; ext/MOIExt.jl:417-566).  `ext/ClusteredLowRankHIPMOIExt.jl` wraps it in an `Optimizer` for JuMP.
"""
function optimize!(opt; device::Integer=0)
    opts = Dict{Symbol,Any}(k => v for (k, v) in opt.options)      # (save_settings included: solvesdp raises for what the device loop cannot do)
    status, dualsol, primalsol, t, e = solvesdp(opt.problem; device=device, opts...)
    opt.optimized = true
    opt.result_data[:primalsol] = primalsol
    opt.result_data[:dualsol] = dualsol
    opt.result_data[:status] = status
    opt.result_data[:errorcode] = e
    opt.result_data[:solve_time] = t
    return status, dualsol, primalsol, t, e
end

# `ClusteredLowRankHIP.Optimizer`: the MOI optimizer type lives in the package extension (ext/ClusteredLowRankHIPMOIExt.jl, loaded with
# MathOptInterface), which registers it here; JuMP takes any callable that returns an optimizer: `GenericModel{BigFloat}(ClusteredLowRankHIP.Optimizer)`
const OPTIMIZER_TYPE = Ref{Any}(nothing)
function Optimizer(; kwargs...)
    OPTIMIZER_TYPE[] === nothing && error("ClusteredLowRankHIP.Optimizer needs MathOptInterface (or JuMP) to be loaded")
    return OPTIMIZER_TYPE[](; kwargs...)
end

end # module
