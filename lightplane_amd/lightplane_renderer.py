"""Alias of the reference's ``lightplane/lightplane_renderer.py`` import path (re-exports only; the code lives in ``renderer.py``)."""
from .renderer import LightplaneFunction, lightplane_renderer  # noqa: F401

__all__ = ["LightplaneFunction", "lightplane_renderer"]
