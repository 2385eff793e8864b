"""DCTPadding lives in pipeline/geometry.py; this module keeps the reference's import path."""
from .geometry import DCTPadding  # noqa: F401
