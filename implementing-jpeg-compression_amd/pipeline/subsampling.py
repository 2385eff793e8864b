"""SubSampling lives in pipeline/geometry.py; this module keeps the reference's import path."""
from .geometry import SubSampling  # noqa: F401
