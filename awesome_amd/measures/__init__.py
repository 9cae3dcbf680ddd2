from .losses import (SE, WeightedLoss, UnariesWeightedLoss, UnariesConversionLoss, criterion_targets, MIOU, AwesomeImageLoss,  # noqa: F401
                     AwesomeLoss, AwesomeImageLossJoint, AwesomeLossJoint, GradientPenaltyLoss, RegularizerLoss, TV, FBMSJointLoss,
                     criterion_to_desc, joint_criterion_form)
