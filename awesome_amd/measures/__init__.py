from .losses import SE, UnariesWeightedLoss, MIOU, AwesomeImageLoss, AwesomeLoss, FBMSJointLoss, criterion_to_desc  # noqa: F401
