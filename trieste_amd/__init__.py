"""trieste_amd: MI355X-native GP-posterior + batch-acquisition engine behind trieste's protocols.

Hot path only: GaussianProcessRegression.update/predict/predict_joint,
ExpectedImprovement, BatchMonteCarloExpectedImprovement, decoupled Thompson trajectories and the
candidate sweep + arg-max of EfficientGlobalOptimization / DiscreteThompsonSampling -- as
hand-written HIP for gfx950 behind a C-ABI (include/tgp.h).  No CPU fallback.
"""
__version__ = "0.1.0"

from .rng import set_seed  # noqa: E402,F401  (the analogue of tf.random.set_seed for the host-side draws)
