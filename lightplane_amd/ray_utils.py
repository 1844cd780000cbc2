"""Alias of the reference's ``lightplane/ray_utils.py`` import path (re-exports only; the code lives in ``rays.py``)."""
from .rays import Rays, calc_harmonic_embedding, calc_harmonic_embedding_dim, jitter_near_far  # noqa: F401

__all__ = ["Rays", "calc_harmonic_embedding", "calc_harmonic_embedding_dim", "jitter_near_far"]
