from .convex_net import ConvexNet, ConvexNextNet  # noqa: F401
from .diffeomorphism_net import ConvexDiffeomorphismNet, NormalBlock, NormalizingFlow1D, WNLinear, WNScale  # noqa: F401
from .wrapper_module import ForwardModule, WrapperModule  # noqa: F401
