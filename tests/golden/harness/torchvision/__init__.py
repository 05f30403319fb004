"""Minimal stand-in for torchvision, used ONLY in the dev container to import
/root/reference (torchvision is not installed here). Test tooling, not product.
Provides: models.VGG / models.alexnet, transforms.* names, datasets.ImageFolder."""
from . import models, transforms, datasets  # noqa: F401


def get_image_backend():
    return "PIL"
