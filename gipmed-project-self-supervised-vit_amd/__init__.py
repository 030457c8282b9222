"""MI355X-native ViT + DINO multi-crop training hot path.

Import through the ``gipvit`` alias at the repo root (the directory name carries
hyphens): ``import gipvit; from gipvit import ops``.
"""
__version__ = "0.1.0"
