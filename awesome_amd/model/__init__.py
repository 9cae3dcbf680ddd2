from .convex_net import ConvexNet, ConvexNextNet  # noqa: F401
from .diffeomorphism_net import (ConvexDiffeomorphismNet, NormalBlock, NormalizingFlow1D, ResidualBlock1D, SimpleBackbone, SimpleResnet, WNLinear,  # noqa: F401
                                 WNScale)
from .pretrainable_module import PretrainableModule  # noqa: F401
from .wrapper_module import ConvSegStandIn, ForwardModule, WrapperModule  # noqa: F401
from .path_connected_net import NoisyPathConnectedNet, PathConnectedNet, real_nvp_path_connected_net  # noqa: F401
from .fc_net import FCNet  # noqa: F401
from .zoo import Zoo  # noqa: F401
from .encoded_mlp import FourierFeatureNet, SineLayerNet  # noqa: F401,E402
from .symmetric_net import RotationSymmetricNet, polar_symmetry_features  # noqa: F401,E402
from .star_net import StarShapedNet  # noqa: F401,E402
