"""`ModelFactory` / `ModelType` with the reference's surface (reference model/model_factory.py:1-37)."""
from enum import Enum

import torch

from .autoregressive_transformer import AutoRegressiveTransformer


class ModelType(Enum):
    MULTI_CLASSES = "multi_classes"


def strip_prefixes(state_dict):
    """DDP / torch.compile checkpoints carry `module.` / `module._orig_mod.` prefixes (reference :27-35)."""
    out = {}
    for k, v in state_dict.items():
        if k.startswith("module._orig_mod."):
            out[k.replace("module._orig_mod.", "")] = v
        elif k.startswith("module."):
            out[k.replace("module.", "")] = v
        else:
            out[k] = v
    return out


class ModelFactory:
    def load_model(self, model_name, model_path, device, model_config=None):
        """Working version of the reference's (broken, unused) `load_model` (:10-13): checkpoint -> model."""
        ckpt = torch.load(model_path, map_location="cpu")
        if model_config is None:
            raise ValueError("load_model needs the model_config the checkpoint was trained with")
        return self.create_model(model_name, model_config, device, state_dict=ckpt["model_state_dict"])

    def create_model(self, model_name, model_config, device, state_dict=None):
        """`model_name` is ignored exactly as in the reference (:22): every config builds an AutoRegressiveTransformer.
        Extra keys in `model_config` ("model_name", "train_config", "state_dict", ...) are tolerated (reference **kwargs)."""
        model = AutoRegressiveTransformer(**model_config).to(device)
        if state_dict:
            print("Loading state dict")
            model.load_state_dict(strip_prefixes(state_dict), strict=False)
        return model, ModelType.MULTI_CLASSES

    build = create_model          # BASELINE.json's wording
