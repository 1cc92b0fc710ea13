# ClusteredLowRankHIP.jl -- thin ccall shim binding the reference's hot path to libclrs_hip.so.
#
# NOT a copy of the reference: it depends on ClusteredLowRankSolver.jl, converts its `ClusteredLowRankSDP`
# into the C description once per solve, and provides drop-in methods with the reference's names and
# argument meaning for the two functions on the path:
#
#     compute_T_decomposition!(sdp, S, A_Y, X_inv, Y, ..., cs_map; prec)        (src/solver.jl:1229-1287)
#     the "solve system" stage of compute_search_direction!                     (src/solver.jl:1527-1582)
#
# Host orchestration stays in Julia (Arb); at the boundary the iterates are rounded to Float64 column-major
# arrays (`Float64.(...)`), the results come back as Float64 and are written into the caller's Arb buffers.
# This file cannot be executed in the build container (no Julia there); INTEGRATION.md is the contract.
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

"""Device context + the host arrays that back the description (kept alive for the lifetime of the context)."""
mutable struct HipContext
    handle::Ptr{Cvoid}
    keep::Vector{Any}
    block_off::Vector{Int}      # offsets of the blocks (j,l) in the xy layout
    block_n::Vector{Int}
    jl::Vector{Tuple{Int,Int}}  # (j,l) of every block in description order
    cluster_off::Vector{Int}
    S_off::Vector{Int}
    term_map::Vector{NTuple{5,Int}}   # (j,l,r,s,idx-in-A_Y[j][l][r,s]) of every term, in term order
    n_free::Int
end

check(code::Integer) = code < 0 ? error("clrs-hip: " * unsafe_string(ccall((:clrs_strerror, libclrs[]), Cstring, (Cint,), code)) *
                                        ": " * unsafe_string(ccall((:clrs_last_error, libclrs[]), Cstring, ()))) : Int(code)

f64(x) = Float64(Arblib.midref(x))

"""
    HipContext(sdp, cs_map; device=0)

Replaces `precompute_matrices_bilinear_pairings` (src/solver.jl:985-1059) and the preallocation block
(src/solver.jl:298-317): flattens `sdp.A[j][l][r,s][p]` (cluster-local constraint indices through `cs_map[j]`,
0-based) and `sdp.B[j]` into `clrs_sdp_desc` and creates the device context.
"""
function HipContext(sdp::CLRS.ClusteredLowRankSDP, cs_map; device::Integer=0)
    J = length(sdp.A)
    cluster_P = Int32[size(sdp.c[j], 1) for j in 1:J]
    N = size(sdp.B[1], 2)
    Bflat = Float64[]
    for j in 1:J, k in 1:N, p in 1:cluster_P[j]
        push!(Bflat, f64(sdp.B[j][p, k]))
    end
    bc = Int32[]; bm = Int32[]; bd = Int32[]; bk = Int32[]
    term_ptr = Int64[0]; tp = Int32[]; tr = Int32[]; ts = Int32[]; tk = Int32[]; tl = Float64[]
    tvp = Int64[0]; tvs = Float64[]; tws = Float64[]
    dense_ptr = Int64[0]; dp = Int32[]; dAp = Int64[0]; dA = Float64[]
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
                append!(dA, [f64(M[a, b]) for b in 1:delta for a in 1:delta])
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
            for (p0, r0, s0, k0) in items
                A = Ajl[r0+1, s0+1][inv_cs[p0+1]]
                delta = length(A.vs[k0+1])
                push!(tp, p0); push!(tr, r0); push!(ts, s0); push!(tk, k0); push!(tl, f64(A.lambda[k0+1]))
                append!(tvs, f64.(A.vs[k0+1])); append!(tws, f64.(A.ws[k0+1]))     # Matrix(::LowRankMat) = sum lam * vs * ws' (src/interface.jl:798-800)
                push!(tvp, length(tvs))
                # position of this term's pairing in A_Y[j][l][r,s] (src/solver.jl:1152-1170): running index per (r,s)
                push!(term_map, (j, l, r0 + 1, s0 + 1, count(t -> t[1] == j && t[2] == l && t[3] == r0 + 1 && t[4] == s0 + 1, term_map) + 1))
            end
        end
        push!(bm, m); push!(bd, delta); push!(block_n, m * delta)
        push!(term_ptr, length(tp)); push!(dense_ptr, length(dp))
    end
    keep = Any[cluster_P, Bflat, bc, bm, bd, bk, term_ptr, tp, tr, ts, tk, tl, tvp, tvs, tws, dense_ptr, dp, dAp, dA]
    desc = Ref(SdpDesc(J, N, pointer(cluster_P), pointer(Bflat), length(bc), pointer(bc), pointer(bm), pointer(bd), pointer(bk),
                       pointer(term_ptr), pointer(tp), pointer(tr), pointer(ts), pointer(tk), pointer(tl), pointer(tvp),
                       pointer(tvs), pointer(tws), pointer(dense_ptr), pointer(dp), pointer(dAp), pointer(dA)))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve keep check(ccall((:clrs_ctx_create, libclrs[]), Cint, (Ref{SdpDesc}, Cint, Ref{Ptr{Cvoid}}), desc, device, h))
    ctx = HipContext(h[], keep, cumsum([0; block_n .^ 2]), block_n, jl, cumsum([0; Int.(cluster_P)]), cumsum([0; Int.(cluster_P) .^ 2]),
                     term_map, N)
    finalizer(c -> ccall((:clrs_ctx_destroy, libclrs[]), Cvoid, (Ptr{Cvoid},), c.handle), ctx)
    return ctx
end

"""Pack a BlockDiagonal of BlockDiagonals of ArbRefMatrix into the xy layout (Float64, column-major per block)."""
function pack_xy(ctx::HipContext, M)
    out = Vector{Float64}(undef, ctx.block_off[end])
    for (b, (j, l)) in enumerate(ctx.jl)
        blk = M.blocks[j].blocks[l]
        n = ctx.block_n[b]
        @inbounds for c in 1:n, r in 1:n
            out[ctx.block_off[b]+r+(c-1)*n] = f64(blk[r, c])
        end
    end
    return out
end

"""
Drop-in for `compute_T_decomposition!` (src/solver.jl:1229-1287).  Same arguments; the Arb scratch arguments
(`bilinear_pairings_*`, `tempX`, `leftvecs`, ..., `part_r`) are accepted and ignored.  Mutates `S` (-> L_j),
`A_Y`, `LinvB`, `Q` (-> L_Q) exactly like the reference and throws the reference's `SolverFailure`s.
"""
function compute_T_decomposition!(ctx::HipContext, sdp, S, A_Y, X_inv, Y, args...; prec=precision(S[1]))
    Xc, Yf = pack_xy(ctx, X_inv), pack_xy(ctx, Y)
    Sout = Vector{Float64}(undef, ctx.S_off[end])
    AY = Vector{Float64}(undef, length(ctx.term_map))
    check(ccall((:clrs_set_timing, libclrs[]), Cint, (Ptr{Cvoid}, Cint), ctx.handle, 1))
    check(ccall((:clrs_schur_assemble, libclrs[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                ctx.handle, Xc, Yf, Sout, AY))
    for (t, (j, l, r, s, idx)) in enumerate(ctx.term_map)
        A_Y[j][l][r, s][idx, 1] = Arb(AY[t]; prec)
    end
    st = check(ccall((:clrs_schur_factor, libclrs[]), Cint, (Ptr{Cvoid},), ctx.handle))
    J = length(S)
    if 0 < st <= J
        throw(CLRS.SolverFailure("S was not decomposed succesfully in block $st, try again with higher precision. If this occurred in the first iteration, remove linear dependencies in the PSD part of the constraints or turn preprocessing on."))
    elseif st == J + 1
        throw(CLRS.SolverFailure("Q was not decomposed correctly. Try restarting with a higher precision. If this occurred in the first iteration, remove linear dependencies between free variables or turn preprocessing on."))
    end
    # The factors stay on the device for the solves; copy them back only if the caller inspects S / LinvB / Q.
    t = zeros(Float64, 6)
    check(ccall((:clrs_get_timings, libclrs[]), Cint, (Ptr{Cvoid}, Ptr{Float64}), ctx.handle, t))
    return t[1], t[2], t[3], t[4], t[5]
end

"""
Drop-in for the "solve system" stage of `compute_search_direction!` (src/solver.jl:1527-1582):
`dx`, `dy` are overwritten; `rhs_x` is the vector `-d - <A_*, Z>` (`:1518-1523`), `rhs_y` is `p`.
"""
function solve_system!(ctx::HipContext, dx, dy, rhs_x::Vector{Float64}, rhs_y::Vector{Float64}; prec=precision(dx.blocks[1]))
    dxf = Vector{Float64}(undef, ctx.cluster_off[end]); dyf = Vector{Float64}(undef, max(ctx.n_free, 1))
    check(ccall((:clrs_schur_solve, libclrs[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                ctx.handle, rhs_x, rhs_y, dxf, dyf))
    for j in 1:length(dx.blocks), p in 1:size(dx.blocks[j], 1)
        dx.blocks[j][p, 1] = Arb(dxf[ctx.cluster_off[j]+p]; prec)
    end
    for k in 1:ctx.n_free
        dy[k, 1] = Arb(dyf[k]; prec)
    end
    return nothing
end

end # module
