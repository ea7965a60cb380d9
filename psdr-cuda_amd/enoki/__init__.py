"""Minimal `enoki`-compatible array shim on PyTorch(-ROCm).

The reference's Python boundary exchanges Enoki CUDA arrays (reference src/psdr.cpp:41-44,
examples/run_test.py:2-3).  Enoki itself is removed by this build; this shim provides just the
surface the reference's callers use: Float32 / Vector{2,3}f / Matrix4f in the
`enoki.cuda` (detached) and `enoki.cuda_autodiff` (differentiable) flavours, plus detach /
set_requires_gradient / forward / backward / gradient / slices and a few math helpers.
Storage is a torch tensor ([N] for scalars arrays, [N,k] for vectors, [4,4] for matrices);
automatic differentiation is torch autograd.
"""
from ._array import (  # noqa: F401
    ArrayBase, detach, set_requires_gradient, requires_gradient, forward, backward, gradient, set_gradient,
    slices, sqrt, sqr, hmean, hsum, hmax, hmin, squared_norm, norm, normalize, dot, cross, abs, select, isfinite,
    default_device, register_render_node, cuda_eval, cuda_sync, cuda_malloc_trim, zero, full, arange,
)
from . import cuda, cuda_autodiff  # noqa: F401
