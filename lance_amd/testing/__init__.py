"""Synthetic inputs shared by bench.py and the tests (no datasets exist on the box)."""
from .datagen import sift_like  # noqa: F401
