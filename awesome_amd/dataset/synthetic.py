"""Deterministic synthetic inputs for the BASELINE configs.  No files, no network.

The reference's datasets (SISBOSI / FBMS-59, awesome/dataset/*) need data that is not on disk; the hot path only needs
per-image unaries in [0,1] with the reference's convention fg = 0, bg = 1 (notebooks/how_to/convexity.ipynb cell 7,
awesome/model/path_connected_net.py:832-836)."""
from __future__ import annotations

import numpy as np
import torch


def _convex_hull(pts: np.ndarray) -> np.ndarray:
    """Andrew monotone chain; returns hull vertices counter-clockwise."""
    pts = pts[np.lexsort((pts[:, 1], pts[:, 0]))]

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower, upper = [], []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in pts[::-1]:
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return np.asarray(lower[:-1] + upper[:-1])


def convex_blob_mask(size: int = 256, seed: int = 0, n_vertices: int = 8) -> np.ndarray:
    """C2 'convex blob': convex hull of K vertices at sorted random angles, radii U(40,90)/256*size around a centre
    U(96,160)/256*size, rasterised by half-plane tests.  Returns a bool (size,size) mask (True = object)."""
    rng = np.random.RandomState(seed)
    s = size / 256.0
    ang = np.sort(rng.uniform(0.0, 2.0 * np.pi, n_vertices))
    rad = rng.uniform(40.0, 90.0, n_vertices) * s
    cx, cy = rng.uniform(96.0, 160.0, 2) * s
    pts = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1)
    hull = _convex_hull(pts)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    inside = np.ones((size, size), dtype=bool)
    for i in range(len(hull)):
        a, b = hull[i], hull[(i + 1) % len(hull)]
        inside &= ((b[0] - a[0]) * (yy - a[1]) - (b[1] - a[1]) * (xx - a[0])) >= 0.0
    return inside


def convex_blob_unaries(size: int = 256, seed: int = 0) -> torch.Tensor:
    """(size,size) float32 unaries, fg (object) = 0, bg = 1."""
    return torch.from_numpy(1.0 - convex_blob_mask(size, seed).astype(np.float32))


def noisy_blob_unaries(size: int = 256, seed: int = 0, flip_p: float = 0.1, n_squares: int = 3, square: int = 8) -> torch.Tensor:
    """C5 'UNet-logit refinement' input: the C2 blob as pseudo-labels with every pixel flipped w.p. flip_p plus
    n_squares random false-positive squares, turned into synthetic segmentation logits `+-2 + N(0,1)` (object positive);
    unaries = 1 - sigmoid(logit): soft values in (0,1), fg ~ 0 - what a segmentation backbone hands to the prior fit."""
    rng = np.random.RandomState(10_000 + seed)
    m = convex_blob_mask(size, seed)
    lab = m ^ (rng.uniform(size=m.shape) < flip_p)
    s = max(1, int(round(square * size / 256.0)))
    for _ in range(n_squares):
        y, x = rng.randint(0, size - s, 2)
        lab[y:y + s, x:x + s] = True
    logit = np.where(lab, 2.0, -2.0) + rng.normal(size=m.shape)
    return torch.from_numpy((1.0 - 1.0 / (1.0 + np.exp(-logit))).astype(np.float32))


def disc_unaries(h: int, w: int, cy: float, cx: float, r: float) -> torch.Tensor:
    """C1 disc: fg = 0 inside the disc, bg = 1."""
    yy, xx = np.mgrid[0:h, 0:w]
    disc = ((yy - cy) ** 2 + (xx - cx) ** 2) <= r * r
    return torch.from_numpy(1.0 - disc.astype(np.float32))


from .prior_dataset import PriorDataset, prior  # noqa: E402


class SyntheticPriorDataset(PriorDataset):
    """The synthetic inputs in the item format of the reference's prior datasets (AwesomeDataset.__getitem__,
    awesome/dataset/awesome_dataset.py:291-294, wrapped by `@prior()`):

        ((index, prior_state), ((image, features, xy_clean), target))

    `image` (1, S, S) holds segmentation LOGITS whose sigmoid, inverted, is the unaries of the synthetic item (so
    WrapperModule(ForwardModule(), prior, use_segmentation_output_inversion=...) reproduces them like a trained backbone would);
    `features` is an empty placeholder, `xy_clean` the (2, S, S) linspace grid of Transformator.get_positional_matrices, `target`
    (1, S, S) the clean mask in the unaries' convention (object = 0, background = 1: what the inverted segmentation output is
    trained against).  `kind`: 'blob' | 'noisy_blob' | 'disc' (see SyntheticUnariesDataset)."""

    returns_index = False
    training_batch_size = 1

    def __init__(self, n_images: int = 1, size: int = 256, kind: str = "blob", seed0: int = 0, prior_model_type=None,
                 prior_model_args=None, **kwargs):
        super().__init__(prior_model_type=prior_model_type, prior_model_args=prior_model_args)
        self.kind = kind
        if kind == "sequence":
            masks = dumbbell_sequence_masks(int(size), int(n_images), int(seed0))
            frames = [torch.from_numpy(1.0 - m.astype(np.float32)) for m in masks]
            self._inner = SyntheticUnariesDataset(n_images=n_images, size=size, kind="blob", seed0=seed0)
            self._inner.unaries = lambda i: frames[int(i)]
            self._inner.ground_truth = lambda i: frames[int(i)]
        else:
            self._inner = SyntheticUnariesDataset(n_images=n_images, size=size, kind=kind, seed0=seed0)
        self.size = int(size)
        xs = torch.linspace(0, 1, self.size)
        self._xy = torch.stack([xs[None, :].expand(self.size, self.size), xs[:, None].expand(self.size, self.size)], 0).contiguous()

    def __len__(self) -> int:
        return len(self._inner)

    def unaries(self, i: int) -> torch.Tensor:
        return self._inner.unaries(i)

    def ground_truth(self, i: int) -> torch.Tensor:
        return self._inner.ground_truth(i)

    def ground_truth_batch(self, indices) -> torch.Tensor:
        return self._inner.ground_truth_batch(indices)

    # ---- the rest of the reference's dataset contract -----------------------------------------------------------
    def decode_encoding(self, output: torch.Tensor) -> torch.Tensor:
        """AwesomeDataset.decode_encoding (awesome/dataset/awesome_dataset.py:393-412), binary, 2-D: `>= 0.5`."""
        return (output >= 0.5).to(dtype=torch.float32)

    def split_indices(self):
        """(train, validation) indices (awesome_dataset.py:169-171): every synthetic image is a training image - the per-image
        fits have nothing to validate on."""
        return np.arange(len(self)), np.zeros((0,), dtype=np.int64)

    def get_config(self):
        """base_dataset.py:51-60: the constructor arguments."""
        return dict(n_images=len(self), size=self.size, kind=self.kind, returns_index=self.returns_index,
                    training_batch_size=self.training_batch_size)

    @prior()
    def __getitem__(self, i: int):
        un = self._inner.unaries(int(i)).clamp(1e-6, 1 - 1e-6)
        fg_prob = 1.0 - un                                   # probability of "object" = what sigmoid(seg logits) is
        image = torch.log(fg_prob / (1.0 - fg_prob))[None]   # logits; sigmoid(image) = fg_prob
        target = (self._inner.ground_truth(int(i)) > 0.5).float()[None]
        xy = self._xy
        if self.kind == "sequence":
            t = float(i) / float(max(len(self) - 1, 1))
            xy = torch.cat([xy, torch.full((1, self.size, self.size), t)], 0)
        return (image, torch.zeros(1, 1, 1), xy), target


class SyntheticUnariesDataset:
    """Minimal stand-in for the reference's prior datasets on the hot path: item i is `(grid_desc, unaries_i)` where
    unaries follow the reference convention (fg = 0).  `kind`: 'disc' (C1), 'blob' (C2/C3, seed = index + seed0) or
    'noisy_blob' (C5: soft unaries from noisy synthetic logits)."""

    def __init__(self, n_images: int = 1, size: int = 256, kind: str = "blob", seed0: int = 0, **kwargs):
        self.n_images, self.size, self.kind, self.seed0 = int(n_images), int(size), kind, int(seed0)

    def __len__(self) -> int:
        return self.n_images

    def unaries(self, i: int) -> torch.Tensor:
        if self.kind == "disc":
            s = self.size
            return disc_unaries(s, s, s / 2, s / 2, 15.0 * s / 64.0)
        if self.kind == "blob":
            return convex_blob_unaries(self.size, self.seed0 + i)
        if self.kind == "noisy_blob":
            return noisy_blob_unaries(self.size, self.seed0 + i)
        raise ValueError(f"unknown kind {self.kind}")

    def ground_truth(self, i: int) -> torch.Tensor:
        """Clean unaries (fg = 0) of item i - what a refinement is scored against."""
        if self.kind == "noisy_blob":
            return convex_blob_unaries(self.size, self.seed0 + i)
        return (self.unaries(i) > 0.5).float()

    def ground_truth_batch(self, indices) -> torch.Tensor:
        return torch.stack([self.ground_truth(int(i)).reshape(-1) for i in indices])

    def __getitem__(self, i: int):
        return (self.size, self.size), self.unaries(i)

    def batch(self, indices) -> torch.Tensor:
        return torch.stack([self.unaries(int(i)).reshape(-1) for i in indices])


def dumbbell_sequence_masks(size: int = 128, frames: int = 16, seed: int = 0) -> np.ndarray:
    """C4 '(x,y,t) sequence' (SURVEY.md §8d): an object of two discs joined by a thin bar, translating 2 px per frame; in
    every third frame the bar is missing (the object is only connected through time) - the case the path-connectedness
    prior over (x, y, t) exists for.  Returns bool (frames, size, size), True = object."""
    rng = np.random.RandomState(seed)
    s = size / 128.0
    cy, cx = rng.uniform(50.0, 78.0) * s, rng.uniform(40.0, 56.0) * s
    r, d, bar = 14.0 * s, 26.0 * s, max(1.0, 1.5 * s)
    ang = rng.uniform(-0.5, 0.5)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    out = np.zeros((frames, size, size), dtype=bool)
    for t in range(frames):
        ox, oy = cx + 2.0 * t * s, cy + 0.5 * t * s
        ax, ay = ox - d * np.cos(ang), oy - d * np.sin(ang)
        bx, by = ox + d * np.cos(ang), oy + d * np.sin(ang)
        m = ((xx - ax) ** 2 + (yy - ay) ** 2 <= r * r) | ((xx - bx) ** 2 + (yy - by) ** 2 <= r * r)
        if t % 3 != 2:
            # distance to the segment a-b
            px, py = xx - ax, yy - ay
            vx, vy = bx - ax, by - ay
            u = np.clip((px * vx + py * vy) / (vx * vx + vy * vy), 0.0, 1.0)
            m |= (px - u * vx) ** 2 + (py - u * vy) ** 2 <= bar * bar
        out[t] = m
    return out


class SyntheticSequenceDataset:
    """Item i = one (x, y, t) sequence for the spatio-temporal prior: `coords(i)` (3, T*H*W) = linspace(0,1) x linspace(0,1) x
    t/t_max (awesome/dataset/transformator.py:25-61) and `unaries(i)` (T*H*W,) with fg = 0."""

    def __init__(self, n_sequences: int = 1, size: int = 128, frames: int = 16, seed0: int = 0, **kwargs):
        self.n_images, self.size, self.frames, self.seed0 = int(n_sequences), int(size), int(frames), int(seed0)

    def __len__(self) -> int:
        return self.n_images

    def coords(self) -> torch.Tensor:
        s, t = self.size, self.frames
        lin = torch.linspace(0, 1, s)
        ts = torch.arange(t).float() / float(max(t - 1, 1))
        tt, yy, xx = torch.meshgrid(ts, lin, lin, indexing="ij")
        return torch.stack([xx.reshape(-1), yy.reshape(-1), tt.reshape(-1)], 0)

    def unaries(self, i: int) -> torch.Tensor:
        m = dumbbell_sequence_masks(self.size, self.frames, self.seed0 + i)
        return torch.from_numpy(1.0 - m.astype(np.float32)).reshape(-1)

    def batch(self, indices) -> torch.Tensor:
        return torch.stack([self.unaries(int(i)) for i in indices])
