"""Alias of the reference's ``lightplane/splatter_module.py`` import path (re-exports only; the code lives in ``modules.py``)."""
from .modules import LightplaneMLPSplatter, LightplaneSplatter  # noqa: F401

__all__ = ["LightplaneSplatter", "LightplaneMLPSplatter"]
