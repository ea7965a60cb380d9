"""enoki.cuda: detached array types (reference uses enoki::CUDAArray, include/psdr/types.h:17-20)."""
from ._array import _CLASSES

Float32 = _CLASSES[("f", False)]
Vector2f = _CLASSES[("v2", False)]
Vector3f = _CLASSES[("v3", False)]
Vector4f = _CLASSES[("v4", False)]
Matrix4f = _CLASSES[("m4", False)]
Int32 = _CLASSES[("i", False)]
UInt32 = Int32
Mask = _CLASSES[("b", False)]
