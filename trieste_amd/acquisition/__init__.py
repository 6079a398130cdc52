"""Acquisition functions, optimizers, rules and samplers of the hot path."""
from .combination import Map, Product, Reducer, Sum
from .continuous_thompson_sampling import (GreedyContinuousThompsonSampling, ParallelContinuousThompsonSampling,
                                           negate_trajectory_function)
from .entropy import (GIBBON, GibbonAcquisition, MinValueEntropySearch, gibbon_quality_term,
                      gibbon_repulsion_term, min_value_entropy_search)
from .function import (AugmentedExpectedImprovement, BatchMonteCarloExpectedImprovement,
                       ExpectedConstrainedImprovement, ExpectedImprovement,
                       MakePositive, MonteCarloExpectedImprovement, MultipleOptimismNegativeLowerConfidenceBound,
                       NegativeLowerConfidenceBound, NegativePredictiveMean, PredictiveVariance,
                       ProbabilityOfFeasibility, ProbabilityOfImprovement, multiple_optimism_lower_confidence_bound,
                       predictive_variance,
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
from .rule import (AcquisitionRule, AsynchronousGreedy, AsynchronousOptimization, AsynchronousRuleState,
                   DiscreteThompsonSampling, EfficientGlobalOptimization, RandomSampling)
from .sampler import ExactThompsonSampler, GumbelSampler, ThompsonSampler, ThompsonSamplerFromTrajectory
from .trust_region import (BatchTrustRegionBox, BatchTrustRegionState, SingleObjectiveTrustRegionBox, TREGOBox,
                           TURBOBox, UpdatableTrustRegionBox)
from .utils import select_nth_output, split_acquisition_function, split_acquisition_function_calls
