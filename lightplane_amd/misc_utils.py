"""Alias of the reference's ``lightplane/misc_utils.py`` import path (re-exports only; the code lives in ``grids.py``).

SURVEY.md 8(b) lists ``misc_utils.flatten_grid`` / ``unflatten_grid`` as part of the drop-in boundary
(reference ``lightplane/misc_utils.py:25-70``).
"""
from .grids import (  # noqa: F401
    assert_shape,
    check_grid,
    check_grid_and_color_grid,
    flatten_grid,
    if_not_none_else,
    is_in_bounds,
    pad_feature_to_block_size,
    process_and_flatten_grid,
    unflatten_grid,
)

__all__ = [
    "assert_shape", "check_grid", "check_grid_and_color_grid", "flatten_grid", "if_not_none_else", "is_in_bounds",
    "pad_feature_to_block_size", "process_and_flatten_grid", "unflatten_grid",
]
