"""awesome/model/net_factory.py surface: factories that build a PathConnectedNet (only the RealNVP variant every reference
config uses is built; splines / glow are not)."""
from .path_connected_net import init_realnvp, real_nvp_path_connected_net  # noqa: F401
