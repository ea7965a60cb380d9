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


def make_tree_scene(seed=0, n_leaves=600, res=64, spp=0, sppe=0, sppse=16):
    """Stand-in for the reference's `tree` scenario (examples/config.py:90-109; its 24 130-face tree0.obj is not shipped):
    an area light, a seeded procedural tree -- a six-sided trunk plus `n_leaves` free-standing leaf triangles, i.e. almost
    only BOUNDARY edges -- and a ground plane.  Like the reference's harness (run_test.py:56-58, "no_edge": [0, 2]) the
    light and the plane are loaded with enable_edges = False, so every secondary edge belongs to the tree.
    Returns an unconfigured Scene; Mesh[1] is the tree."""
    import numpy as np
    from . import Scene, Mesh, Diffuse, PerspectiveCamera
    from .scene import look_at
    rng = np.random.default_rng(seed)
    sc = Scene()
    sc.opts.width = sc.opts.height = res
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = spp, sppe, sppse, 0
    cam = PerspectiveCamera(30.0, 0.01, 1e4)
    cam.to_world = look_at([0.0, 8.0, 30.0], [0.0, 6.0, 0.0], [0.0, 1.0, 0.0])
    sc.add_sensor(cam)

    def bsdf(rgb, name):
        b = Diffuse(list(rgb)); b.id = name; sc.add_bsdf(b); return b
    black, grey, green = bsdf((0, 0, 0), "light"), bsdf((0.7, 0.7, 0.7), "floor"), bsdf((0.4, 0.55, 0.4), "tree")

    def add(verts, faces, b, edges, emitter=None, mid=""):
        m = Mesh()
        m.set_geometry(np.asarray(verts, dtype=np.float32), np.asarray(faces, dtype=np.int32), fname="<%s>" % mid)
        m.use_face_normals, m.enable_edges, m.id = True, edges, mid
        sc.add_mesh(m, b, emitter)
    # light: a 6 x 6 quad up and to the right, facing the crown
    c, n = np.array([9.0, 17.0, 9.0]), np.array([-9.0, -9.0, -9.0]) / np.sqrt(243.0)
    a = np.cross(n, [0.0, 1.0, 0.0]); a /= np.linalg.norm(a); b_ = np.cross(n, a)
    add([c - 3 * a - 3 * b_, c + 3 * a - 3 * b_, c + 3 * a + 3 * b_, c - 3 * a + 3 * b_], [[0, 1, 2], [0, 2, 3]], black, False,
        emitter=[400.0, 400.0, 400.0], mid="light")
    # tree: trunk (closed six-sided prism) + leaves
    verts, faces = [], []
    for k in range(6):
        ang = 2 * np.pi * k / 6
        verts += [[0.5 * np.cos(ang), 0.0, 0.5 * np.sin(ang)], [0.35 * np.cos(ang), 6.0, 0.35 * np.sin(ang)]]
    for k in range(6):
        i0, i1, j0, j1 = 2 * k, 2 * k + 1, 2 * ((k + 1) % 6), 2 * ((k + 1) % 6) + 1
        faces += [[i0, i1, j1], [i0, j1, j0]]
    for _ in range(n_leaves):
        ctr = np.array([0.0, 8.5, 0.0]) + rng.normal(size=3) * np.array([2.2, 1.6, 2.2])
        d1, d2 = rng.normal(size=3), rng.normal(size=3)
        d1 *= rng.uniform(0.35, 0.7) / np.linalg.norm(d1); d2 *= rng.uniform(0.35, 0.7) / np.linalg.norm(d2)
        i = len(verts)
        verts += [list(ctr), list(ctr + d1), list(ctr + d2)]
        faces.append([i, i + 1, i + 2])
    add(verts, faces, green, True, mid="tree")
    add([[-25, 0, -25], [-25, 0, 25], [25, 0, 25], [25, 0, -25]], [[0, 1, 2], [0, 2, 3]], grey, False, mid="plane")
    sc.finalize()
    return sc
