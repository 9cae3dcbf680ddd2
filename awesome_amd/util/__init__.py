from .prior_cache import PriorCache  # noqa: F401
