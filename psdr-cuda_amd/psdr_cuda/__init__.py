"""psdr_cuda -- drop-in Python surface of uci-rendering/psdr-cuda (reference src/psdr.cpp) on top of
the MI355X-native HIP library (include/psdr_hip.h).  Enoki + OptiX are gone; array values are
objects of the bundled `enoki` shim (torch tensors underneath)."""
import enoki  # noqa: F401  (reference psdr.cpp:42-44 imports enoki at module load)
import enoki.cuda  # noqa: F401
import enoki.cuda_autodiff  # noqa: F401

from .core import (Object, RenderOption, Bitmap1fD, Bitmap3fD, DiscreteDistribution,  # noqa: F401
                   HyperCubeDistribution2f, HyperCubeDistribution3f, RayC, RayD, FrameC, FrameD,
                   SampleRecordC, SampleRecordD, PositionSampleC, PositionSampleD)
from .scene import (BSDF, Diffuse, DiffuseBSDF, RoughConductor, RoughConductorBSDF, Emitter, AreaLight, EnvironmentMap,  # noqa: F401
                    Sensor, PerspectiveCamera, Mesh, Scene, PositionSample, BoundarySegSampleDirect)
from .integrator import Integrator, FieldExtractionIntegrator, DirectIntegrator, PathTracer  # noqa: F401

__all__ = [n for n in dir() if not n.startswith("_")]
