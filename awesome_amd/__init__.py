"""awesome_amd - MI355X-native implementation of the per-image INR fit hot path of jp-schneider/awesome.

The compute lives in csrc/inrfit.hip (C ABI: include/inrfit.h, loaded through ctypes); this package is the host side
that mirrors the reference's plugin surface for that path.  There is no CPU fallback."""
from . import _lib  # noqa: F401
from .icnn import FitResult, Grid, IcnnSpec, fit, forward, loss_grad, miou, pack_masks, pack_state_dict, unpack_params  # noqa: F401
from .prior_bank import PriorBank  # noqa: F401

__all__ = ["IcnnSpec", "Grid", "FitResult", "fit", "forward", "loss_grad", "miou", "pack_masks", "pack_state_dict", "unpack_params", "PriorBank"]
