"""Process-wide switches of the MI355X path (plain module attributes).

``check_inputs``       validate ``rays.grid_idx`` against the grid batch size before launching
                       (one device sync per call, like the reference's ``grid_idx.min()/max()``
                       asserts, lightplane_renderer.py:464-467).  Default on.
``check_finite_grads`` run the reference's post-backward ``isfinite`` asserts
                       (lightplane_renderer.py:719-722; a device sync each).  Default off.
``fused_module_ops``   ``LightplaneRenderer.forward`` computes the harmonic ray embedding + its Linear layer in one HIP
                       kernel and the background / alpha epilogue inside the render kernel (2 launches instead of
                       ~17).  Off: the reference's PyTorch op chain around the functional renderer.  Default on.
``segment_backward``   small Renderer batches (<= 32 768 rays, default decoder shape): the backward sweeps every
                       block of LP_SEG_LEN = 8 samples of a ray (two blocks once the batch has more than ~2 000 rays) in its own workgroup, from running sums the forward saves per block
                       (32 B per ray and block).  Default on.
``segment_forward``    the forward of such a batch marches the segments in parallel too (one workgroup per 128 rays and
                       segment + a combine pass; outputs differ from the single sweep by rounding, ~1e-7).  Default on.
``warn_generic_kernel`` warn (once per shape) when a call falls back to the shape-generic kernels, which
                       are one to two orders of magnitude slower than the MFMA / walk families.  Default on.
``stop_transmittance`` early ray termination of the Renderer (extension, see ``lightplane_renderer``): a wavefront stops
                       marching once every ray's transmittance is below this value.  0 (default) = off, exact.
``arithmetic``         arithmetic of the Renderer backward (``LpRendererArgs.arithmetic``, include/lightplane_hip.h): 0 (default) =
                       two-limb bf16 operands in the dX chains / weight gradients where the kernel family uses them (inside
                       1e-4 of the fp32 reference), 1 = ``LP_ARITH_FP32``, the reference's arithmetic (three limbs, fp32
                       weight-gradient products; shapes outside the tuned family then run the shape-generic fp32 kernels).
``march_order``        how the Renderer BACKWARD deals (ray, sample) pairs to a wavefront (``LpRendererArgs.march_order``): "rays" = 32
                       consecutive rays at one sample (its gradient scatter merges neighbouring rays: image-coherent batches),
                       "samples" = 32 consecutive samples of one ray (merges the samples a ray spends in one cell: batches of
                       unrelated rays, e.g. random training batches -- 3x on the reference's speed benchmark), "auto" (default) =
                       "samples" unless most consecutive rays of the batch are neighbours (directions within 5 %, origins within
                       0.05), decided together with the ``check_inputs`` device sync (no extra sync; "rays" when ``check_inputs`` is off).
"""
import os

check_inputs: bool = os.environ.get("LIGHTPLANE_AMD_CHECK_INPUTS", "1") != "0"
check_finite_grads: bool = os.environ.get("LIGHTPLANE_AMD_CHECK_FINITE", "0") == "1"
fused_module_ops: bool = os.environ.get("LIGHTPLANE_AMD_FUSED_MODULE_OPS", "1") != "0"
segment_backward: bool = os.environ.get("LIGHTPLANE_AMD_SEGMENT_BACKWARD", "1") != "0"
segment_forward: bool = os.environ.get("LIGHTPLANE_AMD_SEGMENT_FORWARD", "1") != "0"
warn_generic_kernel: bool = os.environ.get("LIGHTPLANE_AMD_WARN_GENERIC", "1") != "0"
stop_transmittance: float = float(os.environ.get("LIGHTPLANE_AMD_STOP_TRANSMITTANCE", "0"))
arithmetic: int = int(os.environ.get("LIGHTPLANE_AMD_ARITHMETIC", "0"))
march_order: str = os.environ.get("LIGHTPLANE_AMD_MARCH_ORDER", "auto")
