"""bench.py's constants: peaks (MI355X_MICROARCH.md), the workloads (BASELINE.json configs / north star) and their
algorithmic cost per estimate."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SEED = 0x38BEF07CF9CC549D
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured-achievable)
PEAK_F32_MFMA_TF = 157.3     # MI355X_MICROARCH.md: dense f32 MFMA peak (155 TF measured)
PEAK_BF16_MFMA_TF = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak
PLANE_BYTES = 4                 # bytes per operand-plane element of the batch engine (f16 hi + lo planes; ctx.plane_bytes())
PEAK_VALU_GINST = 1024 * 2.4 / 4   # wave64 vector instructions per ns: 256 CUs x 4 SIMDs, one per 4 cycles, 2.4 GHz (MI355X_MICROARCH.md)

# environment variables that do NOT change which kernels run: bench-harness controls and the RCCL library location
BENCH_ENV_OK = {"MIVI_FORCE_DIST", "MIVI_DIST_MODE", "MIVI_DIST_EAGER", "MIVI_BENCH_SKIP_C3", "MIVI_RCCL_LIB", "MIVI_DIST_PIPELINE"}

WORKLOADS = {
    "ns": dict(family=1, d=1024, n_mc=256, target="iso", entropy=0,
               name="north-star: d=1024 full-rank Gaussian family, n_mc=256, target MvNormal(5*1, I), ClosedFormEntropy"),
    "c2": dict(family=0, d=1024, n_mc=256, target="iso", entropy=0,
               name="configs[1]: d=1024 mean-field MvLocationScale, n_mc=256, target MvNormal(5*1, I), ClosedFormEntropy"),
    "ns_dense": dict(family=1, d=1024, n_mc=256, target="dense", entropy=0,
                     name="north-star family, dense-Gaussian target N(5*1, L L'), L = tril(I + 11'/(2d))"),
    "c3": dict(family=1, d=512, n_mc=128, target="logreg", entropy=0, n=1_000_000,
               name="configs[2]: hierarchical LogReg n=1e6, D=512 (511 coefficients + log sigma), full-rank q0=(0, 0.6 I), n_mc=128"),
    "c5": dict(family=0, d=2048, n_mc=64, target="funnel", entropy=3,
               name="configs[4] per-GPU shard: funnel d=2048 + Stacked bijector, mean-field, STL, 64 samples per GPU"),
    "ns_stl": dict(family=1, d=1024, n_mc=256, target="iso", entropy=3,
                   name="north-star family, StickingTheLandingEntropy (adds the C^-T eps solve)"),
}


def algorithmic_cost(w):
    """SURVEY.md 8(d) per-estimate figures (s = 4 bytes): bytes and flops of one estimate."""
    d, M, s = w["d"], w["n_mc"], 4
    if w["family"] == 0:
        return dict(bytes=4 * d * M * s + 4 * d * s, flops=6 * d * M)
    return dict(bytes=(d * (d + 1) // 2) * s + d * d * s + 4 * d * M * s + 2 * d * s, flops=2 * d * d * M)


for _k, _v in WORKLOADS.items():
    _v["key"] = _k


