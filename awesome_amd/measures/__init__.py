from .losses import SE, UnariesWeightedLoss, UnariesConversionLoss, criterion_targets, MIOU, AwesomeImageLoss, AwesomeLoss, FBMSJointLoss, criterion_to_desc  # noqa: F401
