"""Evaluation and export after the fit: the reference's `get_result` / `split_model_result` / `save_result_mask`
(awesome/run/functions.py:2111-2151, 2432-2487, 2315-2361) with the same names, arguments and outputs, for datasets that follow the
reference's item protocol (`((key, prior_state), (inputs, target))` with `@prior()`), plus the batched on-device form:

  evaluate_dataset   every image's prior logits stay on the device; threshold, fg-IoU (`inrfit_miou`, integer counts) and the masks
                     bit-packed 64 pixels per word (`inrfit_pack_masks`) in ONE pass each; only the packed words (H*W/8 bytes per
                     image) come back and are written as 1-bit PNGs - no float image ever crosses PCIe.

PNG files are written with zlib (8-bit and 1-bit grayscale, filter 0): the reference uses cv2.imwrite on the combined object mask;
the pixel values are identical (object k = value k, background 0)."""
from __future__ import annotations

import os
import struct
import zlib
from typing import Any, Dict, Optional, Sequence

import numpy as np
import torch

from ..dataset.prior_dataset import PriorManager
from ..model.pretrainable_module import decompose_training_item


class MissingGroundTruthError(ValueError):
    pass


def _apply_deep(x: Any, fn):
    if isinstance(x, torch.Tensor):
        return fn(x)
    if isinstance(x, (tuple, list)):
        return type(x)(_apply_deep(v, fn) for v in x)
    if isinstance(x, dict):
        return {k: _apply_deep(v, fn) for k, v in x.items()}
    return x


def get_result(model: torch.nn.Module, dataloader: Any, index: int, model_gets_targets: bool = False,
               raise_on_missing_ground_truth: bool = False):
    """run/functions.py:2111-2151: one image through the model in eval mode under its own prior state (PriorManager swap);
    returns (res, ground_truth, image, fg, bg) with `res` on the CPU.  (`fg` / `bg` are the scribble masks of the convexity
    datasets; the dense-unaries datasets of this path have none.)"""
    device = next(model.parameters()).device
    inputs, labels, _, prior_state = decompose_training_item(dataloader[index], dataloader)
    image = inputs[0] if isinstance(inputs, (tuple, list)) else inputs
    if labels is None and raise_on_missing_ground_truth:
        raise MissingGroundTruthError("No ground truth available, can't evaluate")
    was_training = model.training
    with torch.no_grad():
        model.train(False)
        _input_d = _apply_deep(inputs, lambda x: x[None, ...].to(device=device))
        # store_device = cpu: an evaluation-only swap must not turn the cache's entries into device tensors (a manager built
        # without store_device clears the cache's one, as in the reference; a cache saved afterwards would carry CUDA tensors)
        with PriorManager(model, prior_state, getattr(dataloader, "__prior_cache__", None), store_device=torch.device("cpu"),
                          training=False):
            kw = {}
            if model_gets_targets:
                kw["targets"] = _apply_deep(labels, lambda x: x[None, ...].to(device=device))
            res = model(*_input_d, **kw) if isinstance(_input_d, (tuple, list)) else model(_input_d, **kw)
            res = _apply_deep(res, lambda x: x.detach().cpu())
    if getattr(dataloader, "image_channel_format", None) == "bgr":
        image = image[[2, 1, 0], ...]
    model.train(was_training)
    return res, labels, image, None, None


def split_model_result(res: Any, model, dataloader, image: torch.Tensor, compute_crf: bool = False) -> Dict[str, Any]:
    """run/functions.py:2432-2487 without the CRF branch (out of scope): raw and decoded segmentation / prior."""
    if compute_crf:
        raise NotImplementedError("dense CRF post-processing is outside the hot path (SURVEY.md §8)")
    ret: Dict[str, Any] = dict()
    res_pred, res_prior = model.split_model_output(res, additional_data=ret)[0]
    ret["segmentation_raw"], ret["prior_raw"] = res_pred, res_prior
    decode = getattr(dataloader, "decode_encoding", lambda x: x)
    if res_pred is None or res_pred.numel() == 0:
        res_pred = torch.ones(image.shape[1:])
    res_pred = decode(res_pred)
    if res_pred.shape[-2:] != image.shape[1:]:
        res_pred = res_pred.reshape(res_pred.shape[:-2] + image.shape[1:])
    res_pred = res_pred.squeeze()
    if res_pred.dim() == 2:
        res_pred = res_pred[None, ...]
    if res_prior is not None:
        res_prior = decode(res_prior)
        if res_prior.shape != res_pred.shape:
            res_prior = res_prior.reshape(res_pred.shape)
        res_prior = res_prior.squeeze()
        if res_prior.dim() == 2:
            res_prior = res_prior[None, ...]
    ret["segmentation"], ret["prior"] = res_pred, res_prior
    return ret


# ---- PNG (zlib only) -----------------------------------------------------------------------------------------------------------
def _png(path: str, width: int, height: int, bit_depth: int, rows: np.ndarray) -> None:
    """rows: (height, bytes_per_row) uint8 scanlines of a grayscale image (filter type 0 is prepended here)."""
    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    raw = np.concatenate([np.zeros((height, 1), np.uint8), rows], axis=1).tobytes()
    blob = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, bit_depth, 0, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(blob)
    os.replace(tmp, path)


def write_png_gray(path: str, img: np.ndarray) -> None:
    """(H, W) uint8 -> 8-bit grayscale PNG."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    _png(path, img.shape[1], img.shape[0], 8, img)


_BITREV = np.array([int(f"{i:08b}"[::-1], 2) for i in range(256)], dtype=np.uint8)


def save_packed_mask_png(bits: torch.Tensor, height: int, width: int, path: str) -> None:
    """One image's `inrfit_pack_masks` words (bit i of word w = pixel 64 w + i, row-major) -> a 1-bit grayscale PNG (set bit =
    white).  PNG packs pixels most-significant-bit first, the kernel least-significant first: one table lookup per byte."""
    by = bits.detach().cpu().contiguous().view(torch.uint8).numpy().reshape(-1)      # little-endian words: byte b holds pixels 8b..8b+7
    n = height * width
    if width % 8 == 0:
        rows = _BITREV[by[: n // 8]].reshape(height, width // 8)
    else:   # rows do not end on byte boundaries: repack
        px = np.unpackbits(by, bitorder="little")[:n].reshape(height, width)
        rows = np.packbits(px, axis=1, bitorder="big")
    _png(path, width, height, 1, rows)


def combine_object_masks(mask, invert: bool = True) -> np.ndarray:
    """The colour-index image `save_result_mask` writes (run/functions.py:2337-2361): channel k -> value k + 1, later channels over
    earlier ones, background 0; `invert` flips the channels first (the models' masks have foreground = 0).  (The reference's
    'full occlusion' branch compares the values under the new object with the values of the whole image, of which they are a
    subset: it never removes anything, and neither does this.)"""
    if isinstance(mask, torch.Tensor):
        mask = mask.detach().cpu().numpy()
    mask = np.asarray(mask)
    if mask.ndim == 2:
        mask = mask[None, ...]
    if mask.dtype.kind == "f" and not np.all((mask == 0) | (mask == 1)):
        raise ValueError("mask must be binary")
    mask = mask.astype(bool)
    if invert:
        mask = np.logical_not(mask)
    combined = np.zeros(mask.shape[1:], dtype=np.uint8)
    for k in range(mask.shape[0]):
        combined[mask[k]] = k + 1
    return combined


def save_result_mask(mask, path: str, invert: bool = True) -> None:
    """run/functions.py:2315-2361: C x H x W (or H x W) binary channel mask -> one 8-bit PNG, object k = value k."""
    write_png_gray(path, combine_object_masks(mask, invert))


# ---- batched, on the device ----------------------------------------------------------------------------------------------------
def evaluate_dataset(model: torch.nn.Module, dataloader: Any, indices: Optional[Sequence[int]] = None, threshold: float = 0.5,
                     out_dir: Optional[str] = None, use_prior_sigmoid: bool = True) -> Dict[str, Any]:
    """Every image of `indices` through the model under its own prior (like get_result, but the outputs stay on the device), then
    for the whole set at once: fg-IoU of the prior output against the ground truth (`inrfit_miou`) and the binary masks as
    bit-packed words (`inrfit_pack_masks`); with `out_dir` the masks are written as `<index>.png` (1 bit per pixel, white =
    object).  Returns {'indices', 'iou' [n], 'miou', 'bits' [n, H*W/64] int64 on the CPU, 'shape'}."""
    from .. import icnn as K
    device = next(model.parameters()).device
    indices = list(range(len(dataloader))) if indices is None else [int(i) for i in indices]
    outs, gts, shape = [], [], None
    was_training = model.training
    with torch.no_grad():
        model.train(False)
        for i in indices:
            inputs, labels, _, prior_state = decompose_training_item(dataloader[i], dataloader)
            _input_d = _apply_deep(inputs, lambda x: x[None, ...].to(device=device))
            with PriorManager(model, prior_state, getattr(dataloader, "__prior_cache__", None), store_device=torch.device("cpu"),
                              training=False):
                res = model(*_input_d) if isinstance(_input_d, (tuple, list)) else model(_input_d)
            _, prior = model.split_model_output(res)[0]
            prior = prior if prior is not None else model.split_model_output(res)[0][0]
            shape = tuple(prior.shape[-2:])
            outs.append(prior.reshape(-1))
            gts.append(labels.to(device).reshape(-1).float())
        model.train(was_training)
        out = torch.stack(outs)
        if use_prior_sigmoid and not getattr(model, "use_prior_sigmoid", False):
            out = torch.sigmoid(out)
        tgt = torch.stack(gts)
        # outputs AND ground truth follow the reference's convention (foreground -> 0; MIOU(invert=True), awesome/measures/miou.py):
        # object = value at or below the threshold
        iou = K.miou(out, tgt, threshold, 0.5, invert=True)
        bits = K.pack_masks(out, threshold, invert=True).cpu()
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        for k, i in enumerate(indices):
            save_packed_mask_png(bits[k], shape[0], shape[1], os.path.join(out_dir, f"{i}.png"))
    return {"indices": indices, "iou": iou.cpu(), "miou": float(iou.mean()), "bits": bits, "shape": shape}
