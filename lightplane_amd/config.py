"""Process-wide switches of the MI355X path (plain module attributes).

``check_inputs``       validate ``rays.grid_idx`` against the grid batch size before launching
                       (one device sync per call, like the reference's ``grid_idx.min()/max()``
                       asserts, lightplane_renderer.py:464-467).  Default on.
``check_finite_grads`` run the reference's post-backward ``isfinite`` asserts
                       (lightplane_renderer.py:719-722; a device sync each).  Default off.
``grad_replicas``      extra zero-filled copies of ``grad_grid`` the Renderer backward spreads its
                       atomics over (None = automatic: as many as fit in ``grad_replica_bytes``,
                       at most 31).  Same-row fp32 atomics serialise on MI355X; image-coherent
                       rays hit the same plane rows from every workgroup at the same time.
"""
import os

check_inputs: bool = os.environ.get("LIGHTPLANE_AMD_CHECK_INPUTS", "1") != "0"
check_finite_grads: bool = os.environ.get("LIGHTPLANE_AMD_CHECK_FINITE", "0") == "1"
_gr = os.environ.get("LIGHTPLANE_AMD_GRAD_REPLICAS")
grad_replicas = int(_gr) if _gr is not None else None
grad_replica_bytes: int = 64 << 20
