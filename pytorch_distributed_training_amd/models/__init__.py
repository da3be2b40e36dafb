"""Model zoo (reference ``main.py:40`` uses ResNet-50; BASELINE.json adds the rest)."""
from .resnet import ResNet, resnet18, resnet34, resnet50, resnet101, resnet152
from .vit import VisionTransformer, vit_b_16, vit_l_16, vit_s_16, vit_ti_16, vit_tiny_test

_REGISTRY = {
    "resnet18": resnet18, "resnet34": resnet34, "resnet50": resnet50, "resnet101": resnet101,
    "resnet152": resnet152, "vit_b_16": vit_b_16, "vit_ti_16": vit_ti_16, "vit_s_16": vit_s_16,
    "vit_l_16": vit_l_16, "vit_tiny_test": vit_tiny_test,
}


def build_model(name: str, num_classes: int = 1000, **kw):
    try:
        return _REGISTRY[name.lower()](num_classes=num_classes, **kw)
    except KeyError:
        raise ValueError(f"unknown model {name!r}; choose from {sorted(_REGISTRY)}") from None


__all__ = ["ResNet", "VisionTransformer", "build_model", *sorted(_REGISTRY)]
