"""Mirror of the reference package ``src.utils_functions`` for the names on the hot path
(/root/reference/src/utils_functions/__init__.py star-imports the same modules)."""
from .causal_acquisition_functions import AcquisitionQuotient, CausalExpectedImprovement, CandidateGrid  # noqa: F401
from .causal_optimizer import CausalGradientAcquisitionOptimizer  # noqa: F401
from .cost_functions import Cost, total_cost  # noqa: F401
from .utils import find_current_global, find_next_y_point, find_next_y_points, fit_gaussian_process  # noqa: F401
from .graph_functions import (AdditiveSEM, Term, compute_interventions, get_parameter_space, intervene_dict,  # noqa: F401
                              sample_from_model)
