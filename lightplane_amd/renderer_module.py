"""Alias of the reference's ``lightplane/renderer_module.py`` import path (re-export only; the code lives in ``modules.py``)."""
from .modules import LightplaneRenderer  # noqa: F401

__all__ = ["LightplaneRenderer"]
