from .convex_net import ConvexNet, ConvexNextNet  # noqa: F401
