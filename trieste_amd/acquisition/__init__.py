"""Acquisition functions, optimizers, rules and samplers of the hot path.  Components the
survey marks out of scope that were built in round 1 (trust regions, asynchronous rules, builder combinators, entropy
search and its Gumbel min-value sampler, the small builders NegativePredictiveMean / ProbabilityOfFeasibility / MakePositive /
MultipleOptimismNegativeLowerConfidenceBound / PredictiveVariance / ExpectedConstrainedImprovement) are in
:mod:`trieste_amd.extras`; this package exports SURVEY section 8 rows only."""
from .continuous_thompson_sampling import (GreedyContinuousThompsonSampling, ParallelContinuousThompsonSampling,
                                           negate_trajectory_function)
from .function import (AugmentedExpectedImprovement, BatchMonteCarloExpectedImprovement, ExpectedImprovement,
                       MonteCarloExpectedImprovement, NegativeLowerConfidenceBound, ProbabilityOfImprovement,
                       augmented_expected_improvement, batch_monte_carlo_expected_improvement, expected_improvement,
                       monte_carlo_expected_improvement, negative_lower_confidence_bound,
                       probability_below_threshold)
from .greedy_batch import (Fantasizer, LocalPenalization, PenalizedAcquisition, hard_local_penalizer,
                           local_penalizer, soft_local_penalizer)
from .interface import (AcquisitionFunctionBuilder, AcquisitionFunctionClass, GreedyAcquisitionFunctionBuilder,
                        SingleModelAcquisitionBuilder, SingleModelGreedyAcquisitionBuilder,
                        SingleModelVectorizedAcquisitionBuilder, VectorizedAcquisitionFunctionBuilder)
from .optimizer import (FailedOptimizationError, automatic_optimizer_selector, batchify_joint, batchify_vectorize,
                        generate_continuous_optimizer, generate_initial_points, generate_random_search_optimizer,
                        optimize_discrete, sample_from_space)
from .rule import AcquisitionRule, DiscreteThompsonSampling, EfficientGlobalOptimization, RandomSampling
from .sampler import ExactThompsonSampler, ThompsonSampler, ThompsonSamplerFromTrajectory
from .utils import select_nth_output, split_acquisition_function, split_acquisition_function_calls
