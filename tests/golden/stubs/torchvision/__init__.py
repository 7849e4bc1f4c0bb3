"""Minimal stand-in so `/root/reference` modules import in this container (torchvision is not installed).
Only names touched at import time by model/trajectory_model.py and trainer.py are provided; none of them
is executed on the ViT hot path."""
from . import models, transforms, utils  # noqa: F401
