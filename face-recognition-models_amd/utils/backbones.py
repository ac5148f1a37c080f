"""Backbone factory with the reference's signature (main_code/utils/backbones.py:11-31).

Upstream returns a torchvision model with ImageNet weights fetched from the network.  Here
'resnet50' returns the parameter container of the native MI355X engine (torchvision state-dict
names, so reference checkpoints load); weights are torchvision's random init unless `weights`
points at a local state dict.  The other three names the reference accepts have no native engine
(out of the hot-path scope) and say so; unknown names raise ValueError like upstream (:29)."""
import torch

from frx.module import NativeBackbone

from .config import FEATURE_DIM

_NOT_NATIVE = ("resnet18", "efficientnet_b0", "mobilenet_v2")


def get_backbone(backbone_name="resnet18", weights=None):
    if backbone_name == "resnet50":
        bb = NativeBackbone(FEATURE_DIM)
        if weights is not None:
            sd = torch.load(weights, map_location="cpu", weights_only=True)
            sd = {k[len("backbone."):] if k.startswith("backbone.") else k: v for k, v in sd.items()}
            bb.load_state_dict(sd, strict=False)
        return bb
    if backbone_name in _NOT_NATIVE:
        raise NotImplementedError(f"backbone '{backbone_name}' has no MI355X-native engine; use 'resnet50' "
                                  f"(set utils.config.BACKBONE / FR_BACKBONE)")
    raise ValueError(f"Unsupported backbone: {backbone_name}")
