"""EXTRAS -- components OUTSIDE the hot-path scope of SURVEY.md section 8, built in round 1 and frozen since.

SURVEY.md section 2 marks these rows out of scope for this build (row 7: trust regions and asynchronous rules; row 22:
builder combinators and the entropy-search family with its Gumbel sampler; row 5: the small builders of ``builders.py``).  They are host-side orchestration over the same engine-backed
functions, they are tested and they work, so they are kept -- but behind this boundary: nothing in the hot-path
package (``trieste_amd.acquisition``, ``trieste_amd.models``, the loops) imports from here, and no new work goes in.

For convenience the namespace is a superset of ``trieste_amd.acquisition``.
"""
from ..acquisition import *  # noqa: F401,F403
from .async_rules import AsynchronousGreedy, AsynchronousOptimization, AsynchronousRuleState
from .builders import (ExpectedConstrainedImprovement, MakePositive, MultipleOptimismNegativeLowerConfidenceBound,
                       NegativePredictiveMean, PredictiveVariance, ProbabilityOfFeasibility,
                       multiple_optimism_lower_confidence_bound, predictive_variance)
from .combination import Map, Product, Reducer, Sum
from .entropy import (GIBBON, GibbonAcquisition, MinValueEntropySearch, gibbon_quality_term, gibbon_repulsion_term,
                      min_value_entropy_search)
from .samplers import GumbelSampler
from .trust_region import (BatchTrustRegionBox, BatchTrustRegionState, SingleObjectiveTrustRegionBox, TREGOBox, TURBOBox,
                           UpdatableTrustRegionBox)
