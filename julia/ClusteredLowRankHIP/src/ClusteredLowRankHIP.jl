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

const DATA_LIMBS = 2

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

"""
    HipContext(sdp, cs_map; device=0, limbs=5)

Replaces `precompute_matrices_bilinear_pairings` (src/solver.jl:985-1059) and the preallocation block
(src/solver.jl:298-317): flattens `sdp.A[j][l][r,s][p]` (cluster-local constraint indices through `cs_map[j]`,
0-based) and `sdp.B[j]` into `clrs_sdp_desc` and creates the device context.  `limbs = limbs_for(prec)`.
"""
function HipContext(sdp::CLRS.ClusteredLowRankSDP, cs_map; device::Integer=0, limbs::Integer=5)
    J = length(sdp.A)
    DL = limbs == 1 ? 1 : DATA_LIMBS
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
            check(ccall((:clrs_mw_create_ex, libclrs[]), Cint, (Ref{SdpDesc}, Cint, Cint, Cint, Ref{Ptr{Cvoid}}), desc, DL, device, limbs, h))
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

end # module
