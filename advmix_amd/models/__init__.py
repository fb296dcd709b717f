"""Drop-in for the reference's ``lib/models`` package: ``models.<cfg.MODEL.NAME>.get_pose_net``
(tools/train.py:60) and ``models.Unet_generator.UnetGenerator`` (tools/train.py:67)."""
from . import pose_hrnet, pose_resnet, Unet_generator  # noqa: F401
