"""enoki.cuda_autodiff: differentiable array types (reference DiffArray<CUDAArray<T>>)."""
from ._array import _CLASSES

Float32 = _CLASSES[("f", True)]
Vector2f = _CLASSES[("v2", True)]
Vector3f = _CLASSES[("v3", True)]
Vector4f = _CLASSES[("v4", True)]
Matrix4f = _CLASSES[("m4", True)]
Int32 = _CLASSES[("i", False)]
UInt32 = Int32
Mask = _CLASSES[("b", False)]
