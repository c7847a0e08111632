"""Host-side mirror of the reference's ``adapteacher.modeling.GModule`` operator interface for the TTA
path: same class names, constructor signatures, attribute / state-dict names and error
behaviour; every forward runs on the HIP kernels of ``csrc/``."""
from .build_graph import PrototypeComputation  # noqa: F401
from .multi_graph_matching import GA_GM, MGM3_unsup, U_sup  # noqa: F401
