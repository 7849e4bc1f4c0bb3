def _unavailable(*a, **k):
    raise RuntimeError("torchvision stub: resnet/vit_b_16 are out of scope (need network weights)")
resnet18 = vit_b_16 = _unavailable
class ResNet18_Weights:  # noqa: N801
    IMAGENET1K_V1 = None
