from .losses import SE, UnariesWeightedLoss, MIOU, AwesomeImageLoss, FBMSJointLoss, criterion_to_desc  # noqa: F401
