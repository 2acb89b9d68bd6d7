# MIVI.jl -- Julia-side glue that makes libmivi a drop-in for AdvancedVI.jl's RepGradELBO hot path.
#
# UNTESTED IN THIS REPOSITORY'S CI: the build image has no Julia toolchain (SURVEY.md fact 2).  The same
# C ABI is exercised end-to-end from Python (advancedvi.jl_amd/*.py + tests/); this file is the binding a
# maintainer would add on the reference side, see INTEGRATION.md.
#
# Seam (SURVEY.md 8b): `KLMinRepGradDescent`'s objective type is bounded to RepGradELBO, so the plug point is
# the `adtype` argument: `AutoMIVI()` selects more specific methods of `AdvancedVI.init` and
# `AdvancedVI.estimate_gradient!` (src/algorithms/repgradelbo.jl:41-70, 151-177) that call libmivi instead of
# preparing / running an AD backend.  `optimize`, `step`, ClipScale, Optimisers rules stay untouched.
module MIVI

using AdvancedVI, ADTypes, DiffResults, LogDensityProblems, Optimisers, Random, LinearAlgebra
using Distributions: Normal
using AdvancedVI: MvLocationScale, RepGradELBO, ClosedFormEntropy, ClosedFormEntropyZeroGradient,
                  MonteCarloEntropy, StickingTheLandingEntropy, StickingTheLandingEntropyZeroGradient

const libmivi = get(ENV, "LIBMIVI", "libmivi.so")

struct AutoMIVI <: ADTypes.AbstractADType
    device::Int32
end
AutoMIVI() = AutoMIVI(0)

# mivi_config_t (include/mivi.h)
struct MiviConfig
    dtype::Int32; family::Int32; d::Int32; n_mc::Int32; entropy::Int32; device::Int32
    seed::UInt64; m_offset::Int32; m_total::Int32; stream::Ptr{Cvoid}; own_stream::Int32; reserved::Int32
end

entropy_code(::ClosedFormEntropy) = Int32(0)
entropy_code(::ClosedFormEntropyZeroGradient) = Int32(1)
entropy_code(::MonteCarloEntropy) = Int32(2)
entropy_code(::StickingTheLandingEntropy) = Int32(3)
entropy_code(::StickingTheLandingEntropyZeroGradient) = Int32(4)
dtype_code(::Type{Float32}) = Int32(0)
dtype_code(::Type{Float64}) = Int32(1)
family_code(::MvLocationScale{<:Diagonal}) = Int32(0)
family_code(::MvLocationScale) = Int32(1)

mutable struct MIVIState
    problem::Any                 # the (possibly minibatch-conditioned) LogDensityProblem
    T::DataType                  # element type of params (Float32 / Float64)
    ctx::Ptr{Cvoid}
    estimate_idx::UInt64         # replaces the hidden position of `rng`
    cb::Any                      # keeps the @cfunction closure alive
    distributed::Bool            # a communicator is attached (comm_init!): estimates run sharded over the ranks
    dev::Any                     # device scratch for the sharded route: (params, value, grad) pointers or nothing
end

function check(ctx, status)
    status == 0 && return nothing
    msg = unsafe_string(ccall((:mivi_last_error, libmivi), Cstring, (Ptr{Cvoid},), ctx))
    status == 3 && throw(DomainError(msg))                       # non-positive scale diagonal (what `logdet` throws)
    throw(ErrorException("libmivi status $status: $msg"))
end

# Batched LogDensityProblems.logdensity_and_gradient over the columns of Z: the generic plugin route
# (src/mixedad_logdensity.jl:23-34 seam).  Built-in targets would instead call mivi_set_target_*.
function target_callback(user::Ptr{Cvoid}, Zp::Ptr{Cvoid}, d::Int32, M::Int32, ellp::Ptr{Cvoid}, Gp::Ptr{Cvoid})::Int32
    st = unsafe_pointer_to_objref(user)::MIVIState
    T = st.T
    Z = unsafe_wrap(Array, Ptr{T}(Zp), (Int(d), Int(M)))
    ell = unsafe_wrap(Array, Ptr{T}(ellp), (Int(M),))
    G = unsafe_wrap(Array, Ptr{T}(Gp), (Int(d), Int(M)))
    try
        for m in 1:M
            l, g = LogDensityProblems.logdensity_and_gradient(st.problem, view(Z, :, m))
            ell[m] = l
            G[:, m] .= g
        end
        return Int32(0)
    catch
        return Int32(1)
    end
end

function AdvancedVI.init(rng::Random.AbstractRNG, obj::RepGradELBO, ad::AutoMIVI, q::MvLocationScale, prob, params, restructure)
    T = eltype(params)
    # (checked before any native resource exists: nothing to release on this error path)
    LogDensityProblems.capabilities(prob) isa LogDensityProblems.LogDensityOrder{0} &&
        throw(ArgumentError("libmivi has no AD: the target must provide logdensity_and_gradient (wrap it in ADgradient)"))
    cfg = Ref(MiviConfig(dtype_code(T), family_code(q), length(q), obj.n_samples, entropy_code(obj.entropy),
                         ad.device, rand(rng, UInt64), 0, 0, C_NULL, 1, 0))
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    status = ccall((:mivi_create, libmivi), Int32, (Ref{MiviConfig}, Ref{Ptr{Cvoid}}), cfg, ctx)
    status == 0 || error("mivi_create failed with status $status (no HIP device?)")
    st = MIVIState(prob, T, ctx[], UInt64(0), nothing, false, nothing)
    cb = @cfunction(target_callback, Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}))
    st.cb = cb
    check(st.ctx, ccall((:mivi_set_target_callback, libmivi), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Any), st.ctx, cb, C_NULL, st))
    finalizer(s -> ccall((:mivi_destroy, libmivi), Int32, (Ptr{Cvoid},), s.ctx), st)
    return st
end

function AdvancedVI.estimate_gradient!(rng::Random.AbstractRNG, obj::RepGradELBO, ::AutoMIVI,
                                       out::DiffResults.MutableDiffResult, state::MIVIState, params, restructure, args...)
    T = eltype(params)
    value = Ref{T}(zero(T))
    grad = DiffResults.gradient(out)
    status = ccall((:mivi_estimate_gradient_host, libmivi), Int32,
                   (Ptr{Cvoid}, Ptr{T}, UInt64, Ref{T}, Ptr{T}), state.ctx, params, state.estimate_idx, value, grad)
    state.estimate_idx += 1
    # status 2 (non-finite) is NOT thrown here: `step` raises the reference's own ErrorException
    # from `!isfinite(DiffResults.value(grad_buf))` (src/algorithms/common.jl:83-89)
    status == 2 || check(state.ctx, status)
    DiffResults.value!(out, value[])
    return out, state, (elbo = -value[],)
end

function AdvancedVI.estimate_objective(rng::Random.AbstractRNG, obj::RepGradELBO, q::MvLocationScale, prob, ad::AutoMIVI;
                                       n_samples::Int = obj.n_samples)
    params, re = Optimisers.destructure(q)
    st = AdvancedVI.init(rng, RepGradELBO(min(n_samples, 16384); entropy = obj.entropy), ad, q, prob, params, re)
    T = eltype(params)
    value = Ref{T}(zero(T))
    check(st.ctx, ccall((:mivi_estimate_objective_host, libmivi), Int32, (Ptr{Cvoid}, Ptr{T}, UInt64, Int32, Int32, Ref{T}),
                        st.ctx, params, UInt64(0), n_samples, entropy_code(obj.entropy), value))
    return value[]
end

# set_objective_state_problem (src/algorithms/repgradelbo.jl:31-39): SubsampledObjective swaps the minibatch-conditioned
# problem in before every estimate (src/algorithms/subsampledobjective.jl:85-87).  With the host-callback target only the
# Julia-side problem changes; a native logistic-regression target (see `native_logreg!`) re-points its rows instead.
function AdvancedVI.set_objective_state_problem(state::MIVIState, prob)
    state.problem = prob
    if prob isa NativeLogRegBatch
        check(state.ctx, ccall((:mivi_logreg_select_rows, libmivi), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int64, Float64),
                               state.ctx, prob.rows0, length(prob.rows0), prob.likeadj))
    end
    return state
end

# A target whose arithmetic stays on the GPU: hierarchical logistic regression (docs/src/tutorials/subsampling.md:20-46).
# `AdvancedVI.subsample` returns a NativeLogRegBatch (0-based rows + n_data / n), consumed above.
struct NativeLogReg{XT,YT}
    X::XT            # n x p, column-major (Julia's native layout), eltype = the family's eltype
    y::YT            # Vector{UInt8} in {0, 1}
end
struct NativeLogRegBatch
    parent::NativeLogReg
    rows0::Vector{Int64}
    likeadj::Float64
end
LogDensityProblems.dimension(m::NativeLogReg) = size(m.X, 2) + 1
LogDensityProblems.dimension(m::NativeLogRegBatch) = LogDensityProblems.dimension(m.parent)
AdvancedVI.subsample(m::NativeLogReg, idx) = NativeLogRegBatch(m, Int64.(collect(idx)) .- 1, size(m.X, 1) / length(idx))

function native_logreg!(state::MIVIState, m::NativeLogReg; variant::Int = 0, likeadj::Real = 1.0)
    check(state.ctx, ccall((:mivi_set_target_logreg, libmivi), Int32,
                           (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{UInt8}, Int64, Int32, Float64, Int32),
                           state.ctx, m.X, m.y, size(m.X, 1), variant, likeadj, 0))
    return state
end

# gaussian_expectation_gradient_and_hessian! (src/algorithms/gauss_expected_grad_hess.jl:20-60), the inner estimator of
# KLMinWassFwdBwd / KLMinNaturalGradDescent / KLMinSqrtNaturalGradDescent.  That function has no `adtype` to dispatch on,
# so the seam is the problem argument: wrap the target in `MIVITarget(prob, state)` (state from `AdvancedVI.init` above,
# created for a full-rank `q`) and the more specific method below takes the Stein / Price branch on the GPU.
struct MIVITarget{P}
    prob::P
    state::MIVIState
end
LogDensityProblems.dimension(t::MIVITarget) = LogDensityProblems.dimension(t.prob)
LogDensityProblems.capabilities(::Type{<:MIVITarget}) = LogDensityProblems.LogDensityOrder{1}()
LogDensityProblems.logdensity(t::MIVITarget, x) = LogDensityProblems.logdensity(t.prob, x)
LogDensityProblems.logdensity_and_gradient(t::MIVITarget, x) = LogDensityProblems.logdensity_and_gradient(t.prob, x)

function AdvancedVI.gaussian_expectation_gradient_and_hessian!(
    rng::Random.AbstractRNG, q::MvLocationScale{<:LinearAlgebra.AbstractTriangular,<:Normal}, n_samples::Int,
    grad_buf::AbstractVector{T}, hess_buf::AbstractMatrix{T}, t::MIVITarget) where {T<:Real}
    st = t.state
    params, _ = Optimisers.destructure(q)
    logpi = Ref{T}(zero(T))
    check(st.ctx, ccall((:mivi_gauss_expected_grad_hess_host, libmivi), Int32,
                        (Ptr{Cvoid}, Ptr{T}, UInt64, Int32, Ref{T}, Ptr{T}, Ptr{T}),
                        st.ctx, params, st.estimate_idx, n_samples, logpi, grad_buf, hess_buf))   # hess_buf: dense column-major
    st.estimate_idx += 1
    return logpi[], grad_buf, hess_buf
end

# Constrained supports (README.md:76-82, 91-119; docs/src/tutorials/constrained.md:154-196): a Bijectors.Stacked of identity / exp
# blocks is applied by the library around whatever target is set.  `ranges` are the Stacked's UnitRanges (1-based, as Bijectors
# stores them), `kinds[i]` is :identity or :exp.  An empty list removes the constraint.
function set_bijector!(state::MIVIState, ranges::Vector{UnitRange{Int}}, kinds::Vector{Symbol})
    r = Int32[]
    for rg in ranges
        push!(r, Int32(first(rg) - 1)); push!(r, Int32(last(rg)))          # [begin, end) 0-based
    end
    k = Int32[kd === :exp ? 1 : 0 for kd in kinds]
    check(state.ctx, ccall((:mivi_set_bijector_stacked, libmivi), Int32, (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Int32}),
                           state.ctx, Int32(length(kinds)), r, k))
    return state
end

# Multi-GPU (one process per GPU): the collective lives behind the C ABI (RCCL opened by libmivi).  Rank 0 creates the id,
# the host broadcasts its 128 bytes by its own means (MPI.Bcast!, a file, ...), every rank attaches it.  The context must have
# been created with this rank's slice of the sample axis (MiviConfig.m_offset / m_total; see `init_sharded`).
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    status = ccall((:mivi_comm_unique_id, libmivi), Int32, (Ptr{UInt8},), id)
    status == 0 || error("mivi_comm_unique_id failed with status $status (no RCCL?)")
    return id
end
function comm_init!(state::MIVIState, id::Vector{UInt8}, rank::Integer, world::Integer)
    check(state.ctx, ccall((:mivi_comm_init, libmivi), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32), state.ctx, id, Int32(rank), Int32(world)))
    state.distributed = true
    return state
end
comm_destroy!(state::MIVIState) = (check(state.ctx, ccall((:mivi_comm_destroy, libmivi), Int32, (Ptr{Cvoid},), state.ctx)); state.distributed = false; state)

# one sharded estimate on device pointers (params_dev / value_dev / grad_dev live in HBM: AMDGPU.jl arrays or hipMalloc'd buffers);
# every rank passes the SAME estimate_idx and receives the same value and gradient
function estimate_gradient_dist!(state::MIVIState, params_dev::Ptr{Cvoid}, value_dev::Ptr{Cvoid}, grad_dev::Ptr{Cvoid})
    check(state.ctx, ccall((:mivi_estimate_gradient_dist, libmivi), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Ptr{Cvoid}, Ptr{Cvoid}),
                           state.ctx, params_dev, state.estimate_idx, value_dev, grad_dev))
    state.estimate_idx += 1
    check(state.ctx, ccall((:mivi_synchronize, libmivi), Int32, (Ptr{Cvoid},), state.ctx))
    return state
end

# ProximalLocationScaleEntropy on the host arrays works unchanged (src/optimization/proximal_location_scale_entropy.jl);
# the device-resident variant for a parameter vector that lives in HBM is mivi_prox_scale_entropy.

export AutoMIVI, NativeLogReg, native_logreg!, MIVITarget, set_bijector!, comm_unique_id, comm_init!, comm_destroy!, estimate_gradient_dist!
end # module
