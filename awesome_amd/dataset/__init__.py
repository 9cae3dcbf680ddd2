from .synthetic import SyntheticUnariesDataset, convex_blob_mask, convex_blob_unaries, disc_unaries  # noqa: F401
