"""Padding lives in pipeline/geometry.py; this module keeps the reference's import path."""
from .geometry import Padding  # noqa: F401
