"""empty stub: timm is imported but unused by the reference (model/trajectory_model.py:6)."""
