# ClusteredLowRankHIPMOIExt.jl -- JuMP / MathOptInterface surface of the HIP backend (loaded when MathOptInterface is).
#
# `ClusteredLowRankHIP.Optimizer` is the reference's own MOI optimizer (ext/MOIExt.jl: copy_to, attributes, status mapping, result
# getters all stay the reference's code) with ONE method replaced: `MOI.optimize!` (ext/MOIExt.jl:395-407) calls this package's
# `solvesdp` instead of the reference's.  Every other MOI call is forwarded to the wrapped optimizer.
#
#     model = GenericModel{BigFloat}(ClusteredLowRankHIP.Optimizer)      # examples/jump.jl:5 with the backend swapped
module ClusteredLowRankHIPMOIExt

import MathOptInterface as MOI
import ClusteredLowRankSolver
import ClusteredLowRankHIP
using ClusteredLowRankSolver: objvalue

mutable struct Optimizer <: MOI.AbstractOptimizer
    inner::MOI.AbstractOptimizer          # ClusteredLowRankSolver.Optimizer()
    device::Int
    Optimizer(; device::Integer=0) = new(ClusteredLowRankSolver.Optimizer(), Int(device))
end

function __init__()
    ClusteredLowRankHIP.OPTIMIZER_TYPE[] = Optimizer      # (a Ref the parent declares: extensions may not create bindings in their parent)
    return
end

function MOI.optimize!(o::Optimizer)                                # ext/MOIExt.jl:395-407
    opt = o.inner
    status, dualsol, primalsol, t, e = ClusteredLowRankHIP.optimize!(opt; device=o.device)
    pr = opt.problem
    opt.result_data[MOI.SolveTimeSec()] = t
    opt.result_data[MOI.ObjectiveValue()] = objvalue(pr, primalsol)
    opt.result_data[MOI.DualObjectiveValue()] = pr.objective.constant +
        (-1)^(!pr.maximize) * sum(dualsol.x[i][1] * pr.constraints[i].constant for i in eachindex(dualsol.x))
    return
end

# everything else is the reference's optimizer
MOI.is_empty(o::Optimizer) = MOI.is_empty(o.inner)
MOI.empty!(o::Optimizer) = MOI.empty!(o.inner)
MOI.copy_to(o::Optimizer, src::MOI.ModelLike) = MOI.copy_to(o.inner, src)
MOI.supports(o::Optimizer, attr::MOI.AnyAttribute) = MOI.supports(o.inner, attr)
MOI.supports(o::Optimizer, attr::MOI.AnyAttribute, T::Type) = MOI.supports(o.inner, attr, T)
MOI.supports_constraint(o::Optimizer, F::Type{<:MOI.AbstractFunction}, S::Type{<:MOI.AbstractSet}) = MOI.supports_constraint(o.inner, F, S)
MOI.supports_add_constrained_variables(o::Optimizer, S::Type{<:MOI.AbstractVectorSet}) = MOI.supports_add_constrained_variables(o.inner, S)
MOI.get(o::Optimizer, attr::MOI.AnyAttribute) = MOI.get(o.inner, attr)
MOI.get(o::Optimizer, attr::MOI.AnyAttribute, idx) = MOI.get(o.inner, attr, idx)
MOI.set(o::Optimizer, attr::MOI.AnyAttribute, value) = MOI.set(o.inner, attr, value)
MOI.get(o::Optimizer, ::MOI.SolverName) = "ClusteredLowRankSolver (HIP backend)"

end # module
