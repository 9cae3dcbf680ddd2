"""ctypes binding of libinrfit.so (include/inrfit.h).  There is NO CPU fallback: if the HIP library is missing or a
call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("INRFIT_LIB") or os.path.join(HERE, "csrc", "libinrfit.so")  # INRFIT_LIB: A/B builds (tools/)

# enums (include/inrfit.h)
INR_MODEL_ICNN = 1
INR_GRID_SEPARABLE, INR_GRID_EXPLICIT = 0, 1
INR_LOSS_SE, INR_LOSS_BCE = 0, 1
INR_WEIGHT_NONE, INR_WEIGHT_EQUAL, INR_WEIGHT_RATIO, INR_WEIGHT_SSSDMS, INR_WEIGHT_EXPLICIT = 0, 1, 2, 3, 4
INR_OPT_ADAM, INR_OPT_ADAMAX = 0, 1
INR_OPT_HEADER_FLOATS = 8
INRFIT_ABI_VERSION = 7
INR_ACT_RELU, INR_ACT_COS, INR_ACT_SIN = 0, 1, 2
ACT_KINDS = {"relu": INR_ACT_RELU, "cos": INR_ACT_COS, "sin": INR_ACT_SIN}
INR_FLOW_NORMAL_BLOCK, INR_FLOW_SIMPLE = 0, 1

INR_LOSS_EXTERNAL = 2
LOSS_KINDS = {"se": INR_LOSS_SE, "bce": INR_LOSS_BCE, "external": INR_LOSS_EXTERNAL}
WEIGHT_MODES = {"none": INR_WEIGHT_NONE, "equal": INR_WEIGHT_EQUAL, "ratio": INR_WEIGHT_RATIO, "sssdms": INR_WEIGHT_SSSDMS,
                "explicit": INR_WEIGHT_EXPLICIT}
OPT_KINDS = {"adam": INR_OPT_ADAM, "adamax": INR_OPT_ADAMAX}


class InrModelDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_hidden", C.c_int32), ("in_features", C.c_int32), ("n_layers", C.c_int32),
                ("act0", C.c_int32), ("act_omega", C.c_float)]


class InrGridDesc(C.Structure):
    _fields_ = [("mode", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("n_points", C.c_int64),
                ("xs", C.c_void_p), ("ys", C.c_void_p), ("ts", C.c_void_p), ("coords", C.c_void_p),
                ("coords_image_stride", C.c_int64)]


class InrFlowDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("num_coupling", C.c_int32), ("backbone", C.c_int32)]


INR_RNVP_MAX_FLOWS = 32


class InrRnvpDesc(C.Structure):
    _fields_ = [("channels", C.c_int32), ("hidden_units", C.c_int32), ("n_flows", C.c_int32), ("output_fn", C.c_int32),
                ("output_scale", C.c_float), ("vmin", C.c_float * 3), ("vmax", C.c_float * 3), ("new_min", C.c_float),
                ("new_max", C.c_float), ("masks", C.c_uint32 * INR_RNVP_MAX_FLOWS)]


class InrLossDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("weight_mode", C.c_int32), ("ratio", C.c_float), ("c_fg", C.c_float),
                ("c_bg", C.c_float)]


class InrJointLossDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("weight_mode", C.c_int32), ("ratio", C.c_float), ("alpha", C.c_float), ("beta", C.c_float),
                ("clip_penalty", C.c_int32), ("form", C.c_int32), ("prior_kind", C.c_int32), ("prior_weight_mode", C.c_int32),
                ("prior_ratio", C.c_float), ("gamma", C.c_float), ("extra_penalty", C.c_int32), ("n_scribble", C.c_int64),
                ("target_rule", C.c_int32), ("use_noneclass", C.c_int32), ("noneclass", C.c_float)]   # ABI v7 (zero tail = v6 meaning)


JOINT_FBMS, JOINT_AWESOME_IMAGE, JOINT_AWESOME_PIXEL = 0, 1, 2


class InrStarDesc(C.Structure):
    _fields_ = [("n_hidden", C.c_int32)]


class InrOptDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("clamp", C.c_int32), ("plateau", C.c_int32), ("plateau_patience", C.c_int32),
                ("plateau_factor", C.c_float), ("plateau_threshold", C.c_float), ("plateau_min_lr", C.c_float),
                ("plateau_eps", C.c_float), ("freeze_skips", C.c_int32), ("freeze_input", C.c_int32),
                ("logits_at_last_forward", C.c_int32)]


EXPORTS = {
    # name: (restype, argtypes)
    "inrfit_query": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "inrfit_build_info": (C.c_char_p, []),
    "inrfit_slabs_per_image": (C.c_int, [C.c_int64, C.c_int]),
    "inrfit_debug_set_slab_base": (C.c_int, [C.c_int]),
    "inrfit_supported": (C.c_int, [C.POINTER(InrModelDesc)]),
    "inrfit_param_count": (C.c_int64, [C.POINTER(InrModelDesc)]),
    "inrfit_opt_state_floats": (C.c_int64, [C.POINTER(InrModelDesc)]),
    "inrfit_workspace_bytes": (C.c_int64, [C.POINTER(InrModelDesc), C.POINTER(InrGridDesc), C.c_int]),
    "inrfit_forward": (C.c_int, [C.POINTER(InrModelDesc), C.c_void_p, C.POINTER(InrGridDesc), C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_loss_grad": (C.c_int, [C.POINTER(InrModelDesc), C.c_void_p, C.POINTER(InrGridDesc), C.c_void_p,
                                   C.POINTER(InrLossDesc), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_void_p]),
    "inrfit_backward": (C.c_int, [C.POINTER(InrModelDesc), C.c_void_p, C.POINTER(InrGridDesc), C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_step_only": (C.c_int, [C.POINTER(InrModelDesc), C.c_void_p, C.POINTER(InrGridDesc), C.c_void_p,
                                   C.POINTER(InrLossDesc), C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_mfma_stream": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_void_p]),
    "inrfit_flow_param_count": (C.c_int64, [C.POINTER(InrFlowDesc)]),
    "inrfit_cdn_workspace_bytes": (C.c_int64, [C.POINTER(InrModelDesc), C.POINTER(InrFlowDesc), C.POINTER(InrGridDesc), C.c_int]),
    "inrfit_flow_forward": (C.c_int, [C.POINTER(InrFlowDesc), C.c_void_p, C.POINTER(InrGridDesc), C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_flow_backward": (C.c_int, [C.POINTER(InrFlowDesc), C.c_void_p, C.POINTER(InrGridDesc), C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_cdn_forward": (C.c_int, [C.POINTER(InrModelDesc), C.POINTER(InrFlowDesc), C.c_void_p, C.c_void_p,
                                     C.POINTER(InrGridDesc), C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_cdn_loss_grad": (C.c_int, [C.POINTER(InrModelDesc), C.POINTER(InrFlowDesc), C.c_void_p, C.c_void_p,
                                       C.POINTER(InrGridDesc), C.c_void_p, C.POINTER(InrLossDesc), C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_cdn_fit": (C.c_int, [C.POINTER(InrModelDesc), C.POINTER(InrFlowDesc), C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.POINTER(InrGridDesc), C.c_void_p, C.POINTER(InrLossDesc),
                                 C.POINTER(InrOptDesc), C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_rnvp_param_count": (C.c_int64, [C.POINTER(InrRnvpDesc)]),
    "inrfit_pcn_workspace_bytes": (C.c_int64, [C.POINTER(InrModelDesc), C.POINTER(InrRnvpDesc), C.POINTER(InrGridDesc), C.c_int]),
    "inrfit_rnvp_actnorm_init": (C.c_int, [C.POINTER(InrRnvpDesc), C.c_void_p, C.POINTER(InrGridDesc), C.c_int, C.c_void_p,
                                           C.c_int64, C.c_void_p]),
    "inrfit_rnvp_forward": (C.c_int, [C.POINTER(InrRnvpDesc), C.c_void_p, C.POINTER(InrGridDesc), C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_rnvp_inverse": (C.c_int, [C.POINTER(InrRnvpDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_pack_masks": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "inrfit_rnvp_fit_identity": (C.c_int, [C.POINTER(InrRnvpDesc), C.c_void_p, C.c_void_p, C.POINTER(InrGridDesc),
                                           C.POINTER(InrOptDesc), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_void_p]),
    "inrfit_pcn_forward": (C.c_int, [C.POINTER(InrModelDesc), C.POINTER(InrRnvpDesc), C.c_void_p, C.c_void_p,
                                     C.POINTER(InrGridDesc), C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_pcn_loss_grad": (C.c_int, [C.POINTER(InrModelDesc), C.POINTER(InrRnvpDesc), C.c_void_p, C.c_void_p,
                                       C.POINTER(InrGridDesc), C.c_void_p, C.POINTER(InrLossDesc), C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_pcn_fit": (C.c_int, [C.POINTER(InrModelDesc), C.POINTER(InrRnvpDesc), C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.POINTER(InrGridDesc), C.c_void_p, C.POINTER(InrLossDesc),
                                 C.POINTER(InrOptDesc), C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_fit": (C.c_int, [C.POINTER(InrModelDesc), C.c_void_p, C.c_void_p, C.POINTER(InrGridDesc), C.c_void_p,
                             C.POINTER(InrLossDesc), C.POINTER(InrOptDesc), C.c_int, C.c_int, C.c_int, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_joint_loss_workspace_bytes": (C.c_int64, [C.c_int64]),
    "inrfit_joint_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.POINTER(InrJointLossDesc), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_joint_step_workspace_bytes": (C.c_int64, [C.POINTER(InrModelDesc), C.POINTER(InrGridDesc)]),
    "inrfit_joint_step": (C.c_int, [C.POINTER(InrModelDesc), C.c_void_p, C.c_void_p, C.POINTER(InrGridDesc), C.c_void_p, C.c_void_p,
                                    C.POINTER(InrJointLossDesc), C.POINTER(InrOptDesc), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_pcn_joint_step": (C.c_int, [C.POINTER(InrModelDesc), C.POINTER(InrRnvpDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(InrGridDesc), C.c_void_p, C.c_void_p, C.POINTER(InrJointLossDesc),
                                        C.POINTER(InrOptDesc), C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_cdn_joint_step": (C.c_int, [C.POINTER(InrModelDesc), C.POINTER(InrFlowDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(InrGridDesc), C.c_void_p, C.c_void_p, C.POINTER(InrJointLossDesc),
                                        C.POINTER(InrOptDesc), C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_star_param_count": (C.c_int64, [C.POINTER(InrStarDesc)]),
    "inrfit_star_workspace_bytes": (C.c_int64, [C.POINTER(InrStarDesc), C.c_int64]),
    "inrfit_star_forward": (C.c_int, [C.POINTER(InrStarDesc), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_void_p]),
    "inrfit_star_loss_grad": (C.c_int, [C.POINTER(InrStarDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "inrfit_star_fit": (C.c_int, [C.POINTER(InrStarDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_int64, C.POINTER(InrOptDesc), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_void_p]),
    "inrfit_debug_tanh_exp": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "inrfit_miou": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_int, C.c_void_p,
                              C.c_void_p]),
    "inrfit_timing_begin": (C.c_int, [C.c_int]),
    "inrfit_timing_end": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "inrfit_strerror": (C.c_char_p, [C.c_int]),
}

_lib = None


class InrfitError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libinrfit.so; raises if it has not been built (python -m awesome_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise InrfitError(f"{LIB_PATH} not found: build the HIP extension first (python -m awesome_amd.build). "
                          "awesome_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    ver = C.c_int(0)
    lib.inrfit_query(C.byref(ver), None, None)
    # (INRFIT_ABI_ANY=1: kernel A/B tools only - tools/ab.py times an older build's untouched entry points next to this one's)
    if ver.value != INRFIT_ABI_VERSION and os.environ.get("INRFIT_ABI_ANY", "0") != "1":
        raise InrfitError(f"libinrfit ABI {ver.value} != expected {INRFIT_ABI_VERSION}")
    _lib = lib
    return lib


# Debug switch (tests/test_gpu_determinism.py; env INRFIT_POISON=1): every scratch / output buffer the host side allocates
# uninitialised is pre-filled with NaN, so a kernel that reads a byte it (or an earlier kernel of the same call) has not
# written shows up as a NaN result instead of as run-to-run noise.
POISON = os.environ.get("INRFIT_POISON", "0") == "1"


def scratch(*shape, dtype, device):
    """torch.empty for workspaces and kernel outputs (NaN-filled while POISON is set)."""
    import torch
    if POISON and dtype.is_floating_point:
        return torch.full(shape, float("nan"), dtype=dtype, device=device)
    return torch.empty(*shape, dtype=dtype, device=device)


def scratch_like(t):
    return scratch(*t.shape, dtype=t.dtype, device=t.device)


def build_info() -> str:
    """Compiler and code-generation flags the loaded libinrfit.so was built with (inrfit_build_info)."""
    return load().inrfit_build_info().decode()


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().inrfit_strerror(rc)
        raise InrfitError(f"{what} failed: {msg.decode() if msg else rc} ({rc})")
