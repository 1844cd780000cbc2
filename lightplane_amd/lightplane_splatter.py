"""Alias of the reference's ``lightplane/lightplane_splatter.py`` import path (re-exports only; the code lives in ``splatter.py``)."""
from .splatter import (  # noqa: F401
    LightplaneMLPSplatterFunction,
    LightplaneSplatterFunction,
    lightplane_mlp_splatter,
    lightplane_splatter,
)

__all__ = ["LightplaneSplatterFunction", "LightplaneMLPSplatterFunction", "lightplane_splatter", "lightplane_mlp_splatter"]
