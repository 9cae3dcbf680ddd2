from .synthetic import (SyntheticSequenceDataset, SyntheticUnariesDataset, convex_blob_mask, convex_blob_unaries,  # noqa: F401
                        disc_unaries, dumbbell_sequence_masks, noisy_blob_unaries)
from .prior_dataset import PriorDataset, PriorManager, prior  # noqa: F401,E402
from .synthetic import SyntheticPriorDataset  # noqa: F401,E402
