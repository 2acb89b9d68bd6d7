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

using AdvancedVI, ADTypes, AbstractPPL, DiffResults, LogDensityProblems, Optimisers, Random, LinearAlgebra
using Distributions: Normal
using AdvancedVI: MvLocationScale, RepGradELBO, KLMinRepGradDescent, ClosedFormEntropy, ClosedFormEntropyZeroGradient,
                  MonteCarloEntropy, StickingTheLandingEntropy, StickingTheLandingEntropyZeroGradient

const libmivi = get(ENV, "LIBMIVI", "libmivi.so")

# `target_ad`: the backend that differentiates an ORDER-0 target's `logdensity` on the host.  The reference hands `adtype` itself to AD and
# differentiates the whole estimator through `logdensity` (src/algorithms/repgradelbo.jl:50-57); AutoMIVI's estimator is the closed-form
# VJP, so the only derivative left to provide is the target's own, per sample.  Default: ForwardDiff (BASELINE configs[0]; the user loads it).
struct AutoMIVI{A<:Union{Nothing,ADTypes.AbstractADType}} <: ADTypes.AbstractADType
    device::Int32
    target_ad::A
end
AutoMIVI(device::Integer = 0; target_ad = ADTypes.AutoForwardDiff()) = AutoMIVI(Int32(device), target_ad)

# An order-0 problem (only `logdensity`) as an order-1 one: the reference's own AD shim (src/AdvancedVI.jl:47-82 uses the same two
# AbstractPPL calls) applied to `logdensity` alone.  What README.md:168-174 does by hand with LogDensityProblemsAD.ADgradient.
struct ADTarget{P,A}
    prob::P
    ad::A
end
LogDensityProblems.dimension(t::ADTarget) = LogDensityProblems.dimension(t.prob)
LogDensityProblems.capabilities(::Type{<:ADTarget}) = LogDensityProblems.LogDensityOrder{1}()
LogDensityProblems.logdensity(t::ADTarget, x) = LogDensityProblems.logdensity(t.prob, x)
function LogDensityProblems.logdensity_and_gradient(t::ADTarget, x)
    xv = collect(x)                                           # (a column view of Z: docs/src/tutorials/constrained.md:182-184)
    prep = AbstractPPL.prepare(t.ad, Base.Fix1(LogDensityProblems.logdensity, t.prob), xv)
    val, grad = AbstractPPL.value_and_gradient!!(prep, xv)
    return val, copy(grad)                                    # fresh array: callers may mutate it (gauss_expected_grad_hess.jl:51)
end
AdvancedVI.subsample(t::ADTarget, batch) = ADTarget(AdvancedVI.subsample(t.prob, batch), t.ad)

# The capability dispatch of `init` / `set_objective_state_problem` (repgradelbo.jl:31-39, 50-62).
function ad_problem(ad::AutoMIVI, prob; announce::Bool = true)
    capability = LogDensityProblems.capabilities(typeof(prob))
    capability < LogDensityProblems.LogDensityOrder{1}() || return prob
    ad.target_ad === nothing && throw(ArgumentError(
        "The capability of the supplied `LogDensityProblem` $(capability) is less than $(LogDensityProblems.LogDensityOrder{1}()) and " *
        "AutoMIVI(; target_ad=nothing) has no AD backend to differentiate `logdensity`"))
    announce && @info "The capability of the supplied `LogDensityProblem` $(capability) is less than $(LogDensityProblems.LogDensityOrder{1}()). `AdvancedVI` will attempt to directly differentiate through `LogDensityProblems.logdensity`. If this is not intended, please supply a log-density problem with capability at least $(LogDensityProblems.LogDensityOrder{1}())"
    return ADTarget(prob, ad.target_ad)
end

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
    native::Bool                 # the target runs on the device (mivi_set_target_*): the device-resident `optimize` applies
    target_ad::Any               # AutoMIVI's target_ad: re-applied when SubsampledObjective swaps an order-0 problem in
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
    # order 0 (README.md:64-66, bench/benchmarks.jl:39-41): differentiate through `logdensity` on the host, with the reference's @info
    # (checked before any native resource exists: nothing to release on the error path)
    prob = ad_problem(ad, prob)
    cfg = Ref(MiviConfig(dtype_code(T), family_code(q), length(q), obj.n_samples, entropy_code(obj.entropy),
                         ad.device, rand(rng, UInt64), 0, 0, C_NULL, 1, 0))
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    status = ccall((:mivi_create, libmivi), Int32, (Ref{MiviConfig}, Ref{Ptr{Cvoid}}), cfg, ctx)
    status == 0 || error("mivi_create failed with status $status (no HIP device?)")
    st = MIVIState(prob, T, ctx[], UInt64(0), nothing, false, nothing, false, ad.target_ad)
    finalizer(s -> ccall((:mivi_destroy, libmivi), Int32, (Ptr{Cvoid},), s.ctx), st)
    if prob isa NativeTarget          # a target whose arithmetic stays on the GPU: no host callback at all
        set_native_target!(st, prob)
        return st
    end
    cb = @cfunction(target_callback, Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}))
    st.cb = cb
    check(st.ctx, ccall((:mivi_set_target_callback, libmivi), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Any), st.ctx, cb, C_NULL, st))
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
    state.problem = state.target_ad === nothing ? prob : ad_problem(AutoMIVI(0; target_ad = state.target_ad), prob; announce = false)
    if prob isa NativeLogRegBatch
        check(state.ctx, ccall((:mivi_logreg_select_rows, libmivi), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int64, Float64),
                               state.ctx, prob.rows0, length(prob.rows0), prob.likeadj))
    end
    return state
end

# ---- targets that live on the device ---------------------------------------------------------------------------------------------
# Plain LogDensityProblems (so every other AdvancedVI algorithm can use them on the host) that `init(..., ::AutoMIVI, ...)` recognises
# and hands to the library's fused kernels instead of the host callback: MvNormal with diagonal / Cholesky covariance
# (test/models/normal.jl:36-75, bench/benchmarks.jl:43-47) and Neal's funnel under Stacked([log, identity]) (SURVEY.md 8d).
abstract type NativeTarget end
struct NativeDiagNormal{V<:AbstractVector} <: NativeTarget
    mean::V
    std::V
end
struct NativeDenseNormal{V<:AbstractVector,M<:AbstractMatrix} <: NativeTarget
    mean::V
    chol_L::M        # lower Cholesky factor of the covariance
end
struct NativeFunnel <: NativeTarget
    d::Int
    sigma_v::Float64
end
LogDensityProblems.dimension(p::NativeDiagNormal) = length(p.mean)
LogDensityProblems.dimension(p::NativeDenseNormal) = length(p.mean)
LogDensityProblems.dimension(p::NativeFunnel) = p.d
LogDensityProblems.capabilities(::Type{<:NativeTarget}) = LogDensityProblems.LogDensityOrder{1}()
function LogDensityProblems.logdensity_and_gradient(p::NativeDiagNormal, z)
    u = (z .- p.mean) ./ p.std
    return -sum(abs2, u) / 2 - sum(log, p.std) - length(z) * log(2pi) / 2, -u ./ p.std
end
function LogDensityProblems.logdensity_and_gradient(p::NativeDenseNormal, z)
    L = LowerTriangular(p.chol_L)
    w = L \ (z .- p.mean)
    return -sum(abs2, w) / 2 - logdet(L) - length(z) * log(2pi) / 2, -(L' \ w)
end
function LogDensityProblems.logdensity_and_gradient(p::NativeFunnel, eta)
    e1, x, sv2 = eta[1], @view(eta[2:end]), p.sigma_v^2
    n = p.d - 1
    s2 = sum(abs2, x) * exp(-2e1)
    l = -log(p.sigma_v) - log(2pi) / 2 - e1 - e1^2 / (2sv2) - n * log(2pi) / 2 - n * e1 - s2 / 2 + e1
    g = similar(eta)
    g[1] = -1 - e1 / sv2 - n + s2 + 1
    g[2:end] .= .-x .* exp(-2e1)
    return l, g
end
LogDensityProblems.logdensity(p::NativeTarget, z) = first(LogDensityProblems.logdensity_and_gradient(p, z))

# A native target that DECLARES second-order capability: gaussian_expectation_gradient_and_hessian! then takes the sample average of the
# target's Hessians (src/algorithms/gauss_expected_grad_hess.jl:61-83) -- the constant Hessians of the Gaussian targets, the arrow matrix of the
# funnel, one weighted Gram matrix for the logistic regression (csrc/kernels_hess2.hip) -- instead of Stein's identity.  On the device only:
# the host-side `logdensity_gradient_and_hessian` is not provided (the other AdvancedVI algorithms use the wrapped problem's order-1 methods).
struct SecondOrder{P<:NativeTarget} <: NativeTarget
    prob::P
end
LogDensityProblems.dimension(p::SecondOrder) = LogDensityProblems.dimension(p.prob)
LogDensityProblems.capabilities(::Type{<:SecondOrder}) = LogDensityProblems.LogDensityOrder{2}()
LogDensityProblems.logdensity_and_gradient(p::SecondOrder, z) = LogDensityProblems.logdensity_and_gradient(p.prob, z)
set_native_target!(st, p::SecondOrder) = set_native_target!(st, p.prob)

function set_native_target!(st, p::NativeDiagNormal)
    T = st.T
    check(st.ctx, ccall((:mivi_set_target_diag_gauss, libmivi), Int32, (Ptr{Cvoid}, Ptr{T}, Ptr{T}), st.ctx, Vector{T}(p.mean), Vector{T}(p.std)))
    st.native = true
    return st
end
function set_native_target!(st, p::NativeDenseNormal)
    T = st.T
    check(st.ctx, ccall((:mivi_set_target_dense_gauss, libmivi), Int32, (Ptr{Cvoid}, Ptr{T}, Ptr{T}), st.ctx, Vector{T}(p.mean), Matrix{T}(p.chol_L)))
    st.native = true
    return st
end
function set_native_target!(st, p::NativeFunnel)
    check(st.ctx, ccall((:mivi_set_target_funnel, libmivi), Int32, (Ptr{Cvoid}, Float64), st.ctx, p.sigma_v))
    st.native = true
    return st
end

# ---- the device-resident `optimize` ---------------------------------------------------------------------------------------------
# `KLMinRepGradDescent{Obj,AD,...}` carries the AD type as a parameter (src/algorithms/constructors.jl:44-55), so `optimize`
# (src/optimize.jl:42-81) can be specialised on AutoMIVI: with a device-resident target and no callback the whole loop of
# `step` (src/algorithms/common.jl:69-120) -- estimate_gradient!, Optimisers.update!, operator, averager, the isfinite guard -- runs
# inside mivi_optimize_loop with the parameters, optimiser state and running average in HBM: no 4.2 MB parameter upload and 4.2 MB
# gradient download per step (bench.py `also.ns_host_boundary`: ~1 ms per step through mivi_estimate_gradient_host against ~16 us).
# Anything else (a callback, a host-callback target, a rule / operator / averager the loop does not implement, subsampling,
# extra objargs) takes the reference's own `optimize`.
const libhip = get(ENV, "LIBHIP", "libamdhip64.so")
hipcheck(e) = e == 0 || error("HIP runtime error $e")
function dev_alloc(nbytes::Integer)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    hipcheck(ccall((:hipMalloc, libhip), Int32, (Ref{Ptr{Cvoid}}, Csize_t), p, nbytes))
    return p[]
end
dev_free(p::Ptr{Cvoid}) = ccall((:hipFree, libhip), Int32, (Ptr{Cvoid},), p)
h2d(dst::Ptr{Cvoid}, src::Array) = hipcheck(ccall((:hipMemcpy, libhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Int32), dst, src, sizeof(src), 1))
d2h(dst::Array, src::Ptr{Cvoid}) = hipcheck(ccall((:hipMemcpy, libhip), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Int32), dst, src, sizeof(dst), 2))
dev_zero(p::Ptr{Cvoid}, nbytes::Integer) = hipcheck(ccall((:hipMemset, libhip), Int32, (Ptr{Cvoid}, Int32, Csize_t), p, 0, nbytes))

# mivi_loop_t (include/mivi.h)
struct MiviLoop
    rule::Int32; op::Int32; averager::Int32; n_steps::Int32
    eta::Float64; beta1::Float64; beta2::Float64; adam_eps::Float64
    clip_epsilon::Float64; avg_eta::Float64
    opt_state_dev::Ptr{Cvoid}; avg_params_dev::Ptr{Cvoid}
    estimate_idx0::UInt64; t0::Int64
    elbo_dev::Ptr{Cvoid}
end

loop_rule(o::Optimisers.Descent) = (Int32(0), Float64(o.eta), 0.9, 0.999, 1e-8)
loop_rule(o::Optimisers.Adam) = (Int32(1), Float64(o.eta), Float64(o.beta[1]), Float64(o.beta[2]), Float64(o.epsilon))
# DoG / DoWG (src/optimization/rules.jl:17-64) -- the reference's DEFAULT rules: state (x0, v, r) lives on the device as mivi_dog_state_bytes
# ([x0 (n T), padded to 8 bytes; v, r as Float64]); the step size needs two global norms per step, exchanged inside the launch-free loops
loop_rule(::AdvancedVI.DoG) = (Int32(2), 0.0, 0.9, 0.999, 1e-8)
loop_rule(::AdvancedVI.DoWG) = (Int32(3), 0.0, 0.9, 0.999, 1e-8)
# COCOB (src/optimization/rules.jl:66-96): state (L, G, R, theta, x1) as T[5 n] on the device; alpha travels in the loop's `eta` field
loop_rule(o::AdvancedVI.COCOB) = (Int32(4), Float64(o.alpha), 0.9, 0.999, 1e-8)
loop_rule(::Any) = nothing
loop_op(::AdvancedVI.IdentityOperator) = (Int32(0), 0.0)
loop_op(o::AdvancedVI.ClipScale) = (Int32(1), Float64(o.epsilon))
loop_op(::AdvancedVI.ProximalLocationScaleEntropy) = (Int32(2), 0.0)   # KLMinRepGradProxDescent (constructors.jl:122-157); step size from the rule
loop_op(::Any) = nothing
loop_avg(::AdvancedVI.NoAveraging) = (Int32(0), 0.0)
loop_avg(a::AdvancedVI.PolynomialAveraging) = (Int32(1), Float64(a.eta))
loop_avg(::Any) = nothing

const DEVICE_LOOP_CHUNK = 256     # iterations per mivi_optimize_loop call (bounds the work done past a divergence)

function AdvancedVI.optimize(rng::Random.AbstractRNG, alg::KLMinRepGradDescent{<:RepGradELBO,<:AutoMIVI}, max_iter::Int, prob, q_init,
                             objargs...; show_progress::Bool = true, state = nothing, callback = nothing, kwargs...)
    codes = (loop_rule(alg.optimizer), loop_op(alg.operator), loop_avg(alg.averager))
    fast = callback === nothing && isempty(objargs) && prob isa NativeTarget && q_init isa MvLocationScale && all(!isnothing, codes) &&
           !(codes[2][1] == 2 && (codes[1][1] == 1 || codes[1][1] == 4))   # (the proximal operator takes its step size from Descent / DoG / DoWG: proximal_location_scale_entropy.jl:26-42 has no Adam method)
    if !fast   # the reference's own loop (host-driven `step`, this module's estimate_gradient!)
        return invoke(AdvancedVI.optimize, Tuple{Random.AbstractRNG,AdvancedVI.AbstractVariationalAlgorithm,Int,Any,Any,Vararg{Any}},
                      rng, alg, max_iter, prob, q_init, objargs...; show_progress, state, callback, kwargs...)
    end
    state = isnothing(state) ? AdvancedVI.init(rng, alg, q_init, prob) : state
    st = state.obj_st::MIVIState
    T = st.T
    params, re = Optimisers.destructure(state.q)
    n = length(params)
    (rule, eta, b1, b2, aeps), (op, ceps), (avg, aeta) = codes
    # device buffers: parameters, optimiser state (Adam: m; v), running average, per-iteration elbo
    p_dev = dev_alloc(n * sizeof(T)); h2d(p_dev, params)
    isdog = rule == 2 || rule == 3
    dog_bytes = isdog ? Int(ccall((:mivi_dog_state_bytes, libmivi), Int64, (Ptr{Cvoid},), st.ctx)) : 0
    o_dev = rule == 1 ? dev_alloc(2n * sizeof(T)) : (rule == 4 ? dev_alloc(5n * sizeof(T)) : (isdog ? dev_alloc(dog_bytes) : Ptr{Cvoid}(C_NULL)))
    if rule == 4
        L, G, R, th, x1 = state.opt_st.state     # Optimisers.Leaf(COCOB, (L, G, R, theta, x1)): rules.jl:84-86
        h2d(o_dev, Vector{T}(vcat(L, G, R, th, x1)))
    elseif rule == 1
        leaf = state.opt_st                      # Optimisers.Leaf(Adam, (mt, vt, betat)): warm starts keep their moments
        h2d(o_dev, Vector{T}(vcat(leaf.state[1], leaf.state[2])))
    elseif isdog
        x0, v, r = state.opt_st.state            # Optimisers.Leaf(DoG / DoWG, (x0, v, r)): rules.jl:20-22, 49-51
        h2d(o_dev, Vector{T}(x0))
        h2d(o_dev + (dog_bytes - 16), Float64[v, r])
    end
    a_dev = avg == 1 ? dev_alloc(n * sizeof(T)) : Ptr{Cvoid}(C_NULL)
    avg == 1 && h2d(a_dev, Vector{T}(AdvancedVI.value(alg.averager, state.avg_st)))
    e_dev = dev_alloc(DEVICE_LOOP_CHUNK * sizeof(T))
    info_total = NamedTuple[]
    t_done = state.iteration
    try
        done = 0
        while done < max_iter
            k = min(DEVICE_LOOP_CHUNK, max_iter - done)
            loop = Ref(MiviLoop(rule, op, avg, Int32(k), eta, b1, b2, aeps, ceps, aeta, o_dev, a_dev, st.estimate_idx, Int64(t_done), e_dev))
            status = ccall((:mivi_optimize_loop, libmivi), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{MiviLoop}), st.ctx, p_dev, loop)
            status == 2 && throw(ErrorException("The objective value is not finite. This indicates that the optimization run diverged."))
            check(st.ctx, status)
            elbo = Vector{T}(undef, k)
            d2h(elbo, e_dev)
            for i in 1:k
                push!(info_total, (elbo = elbo[i], iteration = done + i))   # the loop index of THIS call (src/optimize.jl:64-68), not the state's counter
            end
            st.estimate_idx += k
            t_done += k
            done += k
        end
        d2h(params, p_dev)
        opt_st = state.opt_st
        if rule == 1
            mv = Vector{T}(undef, 2n)
            d2h(mv, o_dev)
            beta = (T(b1), T(b2))
            opt_st = Optimisers.Leaf(alg.optimizer, (mv[1:n], mv[(n + 1):end], beta .^ (t_done + 1)))
        elseif rule == 4
            cs = Vector{T}(undef, 5n)
            d2h(cs, o_dev)
            opt_st = Optimisers.Leaf(alg.optimizer, ntuple(k -> cs[((k - 1) * n + 1):(k * n)], 5))
        elseif isdog
            vr = Vector{Float64}(undef, 2)
            d2h(vr, o_dev + (dog_bytes - 16))
            opt_st = Optimisers.Leaf(alg.optimizer, (state.opt_st.state[1], T(vr[1]), T(vr[2])))   # x0 never moves
        end
        avg_st = state.avg_st
        if avg == 1
            pa = Vector{T}(undef, n)
            d2h(pa, a_dev)
            avg_st = (pa, t_done + 1)   # (src/optimization/averaging.jl:38-47: (running average, next step number))
        elseif avg == 0
            avg_st = copy(params)
        end
        state = (prob = prob, q = re(params), iteration = t_done, grad_buf = state.grad_buf, opt_st = opt_st, obj_st = st, avg_st = avg_st)
    finally
        dev_free(p_dev); dev_free(e_dev)
        o_dev != C_NULL && dev_free(o_dev)
        a_dev != C_NULL && dev_free(a_dev)
    end
    return AdvancedVI.output(alg, state), map(identity, info_total), state
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
    state.native = true
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
LogDensityProblems.capabilities(::Type{MIVITarget{P}}) where {P} = LogDensityProblems.capabilities(P)   # the wrapped problem's order decides the branch
LogDensityProblems.logdensity(t::MIVITarget, x) = LogDensityProblems.logdensity(t.prob, x)
LogDensityProblems.logdensity_and_gradient(t::MIVITarget, x) = LogDensityProblems.logdensity_and_gradient(t.prob, x)
LogDensityProblems.logdensity_gradient_and_hessian(t::MIVITarget, x) = LogDensityProblems.logdensity_gradient_and_hessian(t.prob, x)

# Second-order plugin (src/algorithms/gauss_expected_grad_hess.jl:61-83): ell, G and the SUM of the Hessians over the columns of Z
function hessian_callback(user::Ptr{Cvoid}, Zp::Ptr{Cvoid}, d::Int32, M::Int32, ellp::Ptr{Cvoid}, Gp::Ptr{Cvoid}, Hp::Ptr{Cvoid})::Int32
    st = unsafe_pointer_to_objref(user)::MIVIState
    T = st.T
    Z = unsafe_wrap(Array, Ptr{T}(Zp), (Int(d), Int(M)))
    ell = unsafe_wrap(Array, Ptr{T}(ellp), (Int(M),))
    G = unsafe_wrap(Array, Ptr{T}(Gp), (Int(d), Int(M)))
    H = unsafe_wrap(Array, Ptr{T}(Hp), (Int(d), Int(d)))
    try
        fill!(H, zero(T))
        for m in 1:M
            l, g, h = LogDensityProblems.logdensity_gradient_and_hessian(st.problem, view(Z, :, m))
            ell[m] = l
            G[:, m] .= g
            H .+= h
        end
        return Int32(0)
    catch
        return Int32(1)
    end
end

function AdvancedVI.gaussian_expectation_gradient_and_hessian!(
    rng::Random.AbstractRNG, q::MvLocationScale{<:LinearAlgebra.AbstractTriangular,<:Normal}, n_samples::Int,
    grad_buf::AbstractVector{T}, hess_buf::AbstractMatrix{T}, t::MIVITarget) where {T<:Real}
    st = t.state
    params, _ = Optimisers.destructure(q)
    logpi = Ref{T}(zero(T))
    if LogDensityProblems.capabilities(typeof(t.prob)) <= LogDensityProblems.LogDensityOrder{1}()   # the reference's branch test (:31-32)
        check(st.ctx, ccall((:mivi_gauss_expected_grad_hess_host, libmivi), Int32,
                            (Ptr{Cvoid}, Ptr{T}, UInt64, Int32, Ref{T}, Ptr{T}, Ptr{T}),
                            st.ctx, params, st.estimate_idx, n_samples, logpi, grad_buf, hess_buf))   # hess_buf: dense column-major
    else   # second-order capability: the sample average of the Hessians (native Gaussian targets: their constant Hessian)
        if !st.native
            hcb = @cfunction(hessian_callback, Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}))
            check(st.ctx, ccall((:mivi_set_target_hess_callback, libmivi), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Any), st.ctx, hcb, st))
        end
        check(st.ctx, ccall((:mivi_gauss_expected_grad_hess2_host, libmivi), Int32,
                            (Ptr{Cvoid}, Ptr{T}, UInt64, Int32, Ref{T}, Ptr{T}, Ptr{T}),
                            st.ctx, params, st.estimate_idx, n_samples, logpi, grad_buf, hess_buf))
    end
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
# been created with this rank's slice of the sample axis (MiviConfig.m_offset / m_total: build the MiviConfig with n_mc = the local share,
# m_offset = its first global column, m_total = the global n_samples before mivi_create).  `comm_enable_p2p!` then maps the peers' exchange
# areas (the exchange written for xGMI, csrc/kernels_p2p.hip) through the communicator; batched estimates: estimate_gradient_dist_n!.
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
comm_enable_p2p!(state::MIVIState) = (check(state.ctx, ccall((:mivi_comm_enable_p2p, libmivi), Int32, (Ptr{Cvoid},), state.ctx)); state)
function estimate_gradient_dist_n!(state::MIVIState, params_dev::Ptr{Cvoid}, count::Integer, value_dev::Ptr{Cvoid}, grad_dev::Ptr{Cvoid})
    check(state.ctx, ccall((:mivi_estimate_gradient_dist_n, libmivi), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Int32, Ptr{Cvoid}, Ptr{Cvoid}),
                           state.ctx, params_dev, state.estimate_idx, Int32(count), value_dev, grad_dev))
    state.estimate_idx += count
    check(state.ctx, ccall((:mivi_synchronize, libmivi), Int32, (Ptr{Cvoid},), state.ctx))
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

# `count` estimates at fixed parameters on device pointers: `estimate_gradient!` called `count` times without an update in between
# (src/algorithms/repgradelbo.jl:151-177).  values_dev: T[count]; grads_dev: T[count * length(params)] or C_NULL (values only).
# Full-rank Float32 with a native Gaussian target runs on libmivi's batch engine (DESIGN.md 3); every estimate equals the single call's.
function estimate_gradient_each!(state::MIVIState, params_dev::Ptr{Cvoid}, count::Integer, values_dev::Ptr{Cvoid}, grads_dev::Ptr{Cvoid}=C_NULL)
    check(state.ctx, ccall((:mivi_estimate_gradient_each, libmivi), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Int32, Ptr{Cvoid}, Ptr{Cvoid}),
                           state.ctx, params_dev, state.estimate_idx, Int32(count), values_dev, grads_dev))
    state.estimate_idx += count
    check(state.ctx, ccall((:mivi_synchronize, libmivi), Int32, (Ptr{Cvoid},), state.ctx))
    return state
end

# ProximalLocationScaleEntropy on the host arrays works unchanged (src/optimization/proximal_location_scale_entropy.jl);
# the device-resident variant for a parameter vector that lives in HBM is mivi_prox_scale_entropy.

export AutoMIVI, SecondOrder, NativeDiagNormal, NativeDenseNormal, NativeFunnel, NativeLogReg, native_logreg!, MIVITarget, set_bijector!, comm_unique_id, comm_init!, comm_enable_p2p!, comm_destroy!, estimate_gradient_dist!, estimate_gradient_dist_n!, estimate_gradient_each!
end # module
