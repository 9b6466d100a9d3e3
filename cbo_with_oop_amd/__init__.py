"""cbo_with_oop_amd: MI355X-native GP posterior update + causal acquisition sweep behind the
ChampiB/CBO_with_OOP API surface.  Importing the package does not touch the GPU;
creating a model does, and fails loudly when libcbo_hip.so or a gfx950 device is missing."""
from .GaussianProcessFactory import GaussianProcessFactory, GaussianProcessType, HipGaussianProcess  # noqa: F401
from .utils_functions import (CandidateGrid, CausalExpectedImprovement, Cost, find_current_global,  # noqa: F401
                              find_next_y_point, total_cost)
from .CBO import CBOAcquisitionPath  # noqa: F401
from .DoCalculus import DoCalculus, do_prior_functions  # noqa: F401

__version__ = "0.1.0"
