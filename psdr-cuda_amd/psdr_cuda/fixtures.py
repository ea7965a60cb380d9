"""Locations of the bundled scene fixtures (psdr-cuda_amd/data, see its README)."""
import os

from ._abi import PKG_ROOT

DATA_DIR = os.path.join(PKG_ROOT, "data")


def scene_path(name):
    p = os.path.join(DATA_DIR, "scenes", name if name.endswith(".xml") else name + ".xml")
    if not os.path.exists(p):
        raise RuntimeError("unknown scene fixture: " + name)
    return p


def make_interior_scene(seed=0, n_objects=10, res=64, spp=4, sppe=0, sppse=0):
    """BASELINE.json config 5 stand-in (none is shipped by the reference): a seeded multi-material
    interior -- the Cornell-box shell with rough-conductor floor/back wall plus `n_objects` bunny_low
    copies (4 968 triangles each, ~50 k for 10) carrying GGX materials with alpha in [0.05, 0.5] or
    diffuse colours.  Returns an unconfigured Scene."""
    import numpy as np
    import torch
    from . import Scene, Mesh, Diffuse, RoughConductor, PerspectiveCamera
    from .scene import look_at, load_obj

    rng = np.random.default_rng(seed)
    sc = Scene()
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, sppe, sppse, 0
    cam = PerspectiveCamera(13.0, 0.1, 1e4)
    cam.to_world = look_at([0, 125, 1000], [0, 124.965, 999.001], [0, 0.999388, -0.0349786])
    sc.add_sensor(cam)

    def bsdf(b, name):
        b.id = name
        sc.add_bsdf(b)
        return b
    white = bsdf(Diffuse([0.95, 0.95, 0.95]), "white")
    red = bsdf(Diffuse([0.9, 0.2, 0.2]), "red")
    green = bsdf(Diffuse([0.2, 0.9, 0.2]), "green")
    black = bsdf(Diffuse([0.0, 0.0, 0.0]), "black")
    metal_floor = bsdf(RoughConductor(0.25, (0.155, 0.117, 0.138), (4.83, 3.12, 2.15)), "metal_floor")
    metal_back = bsdf(RoughConductor(0.1, (0.2, 0.92, 1.1), (3.9, 2.45, 2.14)), "metal_back")
    obj = os.path.join(DATA_DIR, "objects")

    def add(fname, b, xf=None, face_normals=True, emitter=None, mid=""):
        m = Mesh()
        m.load(os.path.join(obj, fname))
        m.use_face_normals = face_normals
        m.id = mid
        if xf is not None:
            m._to_world_raw = torch.as_tensor(xf, dtype=torch.float32, device=m._to_world_raw.device)
        sc.add_mesh(m, b, emitter)
        return m
    t = np.eye(4); t[:3, 3] = [50, 190, 0]
    add("cbox/emitter.obj", black, t, emitter=[20.0, 20.0, 8.0], mid="light")
    add("cbox/floor.obj", metal_floor); add("cbox/ceil.obj", white); add("cbox/wall_back.obj", metal_back)
    add("cbox/wall_left.obj", red); add("cbox/wall_right.obj", green)
    for i in range(n_objects):
        if rng.random() < 0.7:
            b = bsdf(RoughConductor(float(rng.uniform(0.05, 0.5)), tuple(rng.uniform(0.1, 1.5, 3)), tuple(rng.uniform(2.0, 5.0, 3))), "obj%d" % i)
        else:
            b = bsdf(Diffuse(list(rng.uniform(0.2, 0.9, 3))), "obj%d" % i)
        s = rng.uniform(0.3, 0.5)
        ang = rng.uniform(0, 2 * np.pi)
        c, sn = np.cos(ang), np.sin(ang)
        xf = np.array([[s * c, 0, s * sn, rng.uniform(-70, 70)], [0, s, 0, rng.uniform(20, 150)], [-s * sn, 0, s * c, rng.uniform(-60, 150)], [0, 0, 0, 1]])
        add("bunny/bunny_low.obj", b, xf, face_normals=False, mid="bunny%d" % i)
    sc.finalize()
    return sc
