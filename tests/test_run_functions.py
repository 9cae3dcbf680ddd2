"""Evaluation / export mirror (awesome_amd/run/functions.py; reference awesome/run/functions.py:2111-2151, 2315-2361, 2432-2487):
the colour-index mask image against a literal restatement of the reference's loop, and the two PNG writers against an
independent decoder (PIL).  CPU only."""
import numpy as np
import pytest
import torch

PIL = pytest.importorskip("PIL.Image")


def _reference_combined_mask(mask: np.ndarray, invert: bool) -> np.ndarray:
    """save_result_mask's body (run/functions.py:2337-2359) restated line by line, the 'full occlusion' branch included."""
    if mask.ndim == 2:
        mask = mask[None, ...]
    mask = np.transpose(mask, (1, 2, 0)).astype(bool)
    if invert:
        mask = np.logical_not(mask)
    combined = np.zeros(mask.shape[:2], dtype=np.uint8)
    colors = list(range(1, mask.shape[-1] + 1))
    for k in range(mask.shape[-1]):
        obj = mask[..., k].copy()
        existing = combined[obj]
        occluded = set(np.unique(existing)) - set(np.unique(combined))
        for o in occluded:
            obj[existing == o] = 0
        combined[obj] = colors[k]
    return combined


@pytest.mark.parametrize("channels,invert", [(1, True), (3, True), (4, False)])
def test_save_result_mask_matches_the_reference_loop_and_round_trips(tmp_path, channels, invert):
    from awesome_amd.run import combine_object_masks, save_result_mask
    rng = np.random.RandomState(3 + channels)
    H, W = 37, 53
    mask = np.ones((channels, H, W), np.float32)
    for k in range(channels):       # overlapping discs: later objects cover earlier ones, one is fully covered
        cy, cx, r = rng.randint(8, H - 8), rng.randint(8, W - 8), rng.randint(4, 12)
        yy, xx = np.mgrid[0:H, 0:W]
        mask[k][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 0.0          # the models' convention: object = 0
    if channels == 4:
        mask[3] = mask[0] * 1.0                                            # same pixels again -> object 0 fully occluded
    m = mask if invert else 1.0 - mask
    want = _reference_combined_mask(m.copy(), invert)
    np.testing.assert_array_equal(combine_object_masks(torch.from_numpy(m), invert), want)
    p = str(tmp_path / "mask.png")
    save_result_mask(torch.from_numpy(m), p, invert=invert)
    img = np.asarray(PIL.open(p))
    assert img.dtype == np.uint8 and img.shape == (H, W)
    np.testing.assert_array_equal(img, want)
    with pytest.raises(ValueError):
        combine_object_masks(torch.full((1, 4, 4), 0.5))


@pytest.mark.parametrize("H,W", [(16, 64), (9, 40), (7, 13), (1, 1)])
def test_packed_mask_png_is_the_mask(tmp_path, H, W):
    """`inrfit_pack_masks` layout (bit i of 64-bit word w = pixel 64 w + i) -> 1-bit PNG, widths with and without whole bytes."""
    from awesome_amd.run import save_packed_mask_png
    rng = np.random.RandomState(H * 100 + W)
    px = rng.rand(H * W) > 0.5
    words = (H * W + 63) // 64
    pad = np.zeros(words * 64, dtype=np.uint64)
    pad[: H * W] = px
    bits = (pad.reshape(words, 64) << np.arange(64, dtype=np.uint64)).sum(-1, dtype=np.uint64)
    p = str(tmp_path / "m.png")
    save_packed_mask_png(torch.from_numpy(bits.view(np.int64)), H, W, p)
    im = PIL.open(p)
    assert im.mode == "1" and im.size == (W, H)
    np.testing.assert_array_equal(np.asarray(im).astype(bool), px.reshape(H, W))
