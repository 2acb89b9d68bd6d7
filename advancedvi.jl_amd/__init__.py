"""advancedvi.jl_amd: MI355X-native RepGradELBO / ADVI hot path of AdvancedVI.jl behind the
reference's own interface.  Import as `import advancedvi_jl_amd as avi` (shim at the repo root).

Everything numerical is a hand-written HIP kernel in libmivi.so (csrc/, C ABI in include/mivi.h);
this package is the host-side mirror of the reference's operator interface for that path."""
from ._lib import LIB_PATH, MiviError, load as load_library
from .families import MvLocationScale, MeanFieldGaussian, FullRankGaussian, destructure, MEANFIELD, FULLRANK
from .problems import (LogDensityOrder, DiagNormalProblem, DenseNormalProblem, LogRegProblem, FunnelProblem,
                       dimension, capabilities)
from .objectives import (RepGradELBO, RepGradELBOState, ClosedFormEntropy, ClosedFormEntropyZeroGradient,
                         MonteCarloEntropy, StickingTheLandingEntropy, StickingTheLandingEntropyZeroGradient,
                         AutoMIVI, PhiloxRNG, DiffResult, set_objective_state_problem, rand,
                         gaussian_expectation_gradient_and_hessian_)
from . import objectives as _objectives
from . import optimize as _optimize
from .optimize import (KLMinRepGradDescent, KLMinRepGradProxDescent, ADVI, ClipScale, IdentityOperator,
                       ProximalLocationScaleEntropy, Descent, Adam, DoG, DoWG, COCOB, NoAveraging,
                       PolynomialAveraging, optimize, step, output)
from .context import MiviContext
from .problems import subsample, LogRegSubset, FunnelConstrainedProblem, StackedBijector, TransformedProblem, ADgradient
from . import forwarddiff
from .subsampling import (ReshufflingBatchSubsampling, ReshufflingBatchSubsamplingState, SubsampledObjective,
                          SubsampledObjectiveState)
from . import subsampling as _subsampling
from . import distributed


def estimate_objective(*args, **kwargs):
    """Dispatches like the reference: (rng, alg|obj, q, prob) or (alg|obj, q, prob)."""
    head = args[1] if isinstance(args[0], PhiloxRNG) else args[0]
    if isinstance(head, KLMinRepGradDescent):
        return _optimize.estimate_objective(*args, **kwargs)
    if isinstance(head, SubsampledObjective):
        return _subsampling.estimate_objective(*args, **kwargs)
    return _objectives.estimate_objective(*args, **kwargs)


def estimate_gradient_(rng, obj, *args, **kwargs):
    """`estimate_gradient!(rng, obj, adtype, out, state, params, restructure)` for RepGradELBO or SubsampledObjective."""
    if isinstance(obj, SubsampledObjective):
        return _subsampling.estimate_gradient_(rng, obj, *args, **kwargs)
    return _objectives.estimate_gradient_(rng, obj, *args, **kwargs)


def init(*args, **kwargs):
    """init(rng, alg, q_init, prob)  or  init(rng, obj, adtype, q, prob, params, restructure)."""
    if isinstance(args[1], KLMinRepGradDescent):
        return _optimize.init(*args, **kwargs)
    if isinstance(args[1], SubsampledObjective):
        return _subsampling.init(*args, **kwargs)
    return _objectives.init(*args, **kwargs)
