class _T:
    def __init__(self, *a, **k): pass
    def __call__(self, x): return x
Compose = Resize = Grayscale = ToTensor = Normalize = CenterCrop = _T
