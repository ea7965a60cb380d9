"""ctypes mirror of include/psdr_hip.h and the loader of the library that implements it.

``load_hip()`` -> libpsdr_hip.so, the product (HIP/gfx950 kernels).  Fails loudly when the library is
missing; there is NO CPU fallback in the product path.  (The CPU oracle has its own loader in
oracle/oracle.py: test infrastructure, not part of this package.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)                 # psdr-cuda_amd/
REPO_ROOT = os.path.dirname(PKG_ROOT)

TRI_STRIDE = 24
TRIUV_STRIDE = 8
SEDGE_STRIDE = 16
PEDGE_STRIDE = 8
BSDF_STRIDE = 16
EMITTER_F_STRIDE = 8
EMITTER_I_STRIDE = 4
CAM_WORDS = 64
ENV_WORDS = 32
TRI_FACE_NORMALS = 0x40000000

BSDF_DIFFUSE, BSDF_ROUGHCONDUCTOR = 0, 1
SLOT_REFLECTANCE, SLOT_ALPHA_U, SLOT_ALPHA_V, SLOT_ETA, SLOT_K = range(5)
CAM_SAMPLE_TO_CAMERA, CAM_TO_WORLD, CAM_WORLD_TO_SAMPLE, CAM_POS, CAM_DIR, CAM_INV_AREA = 0, 16, 32, 48, 51, 54
INTEGRATOR_DIRECT, INTEGRATOR_PATH, INTEGRATOR_FIELD = 0, 1, 2
FLAG_FUSED, FLAG_WAVEFRONT, FLAG_LITERAL_FORMS, FLAG_KEEP_RECORDS = 1, 2, 4, 8
FIELDS = {"silhouette": 0, "position": 1, "depth": 2, "geoNormal": 3, "shNormal": 4, "uv": 5}

_fp = C.c_void_p  # all table pointers travel as raw addresses


class SceneDesc(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("num_tris", C.c_int32), ("num_meshes", C.c_int32), ("num_bsdfs", C.c_int32), ("num_emitters", C.c_int32),
        ("num_sec_edges", C.c_int32), ("num_prim_edges", C.c_int32), ("num_texels", C.c_int32),
        ("num_guide_cells", C.c_int32),
        ("tri_info", _fp), ("tri_uv", _fp), ("tri_mesh", _fp), ("mesh_bsdf", _fp), ("mesh_emitter", _fp),
        ("bsdf_rec", _fp), ("texels", _fp), ("emitter_f", _fp), ("emitter_i", _fp),
        ("face_cmf", _fp), ("face_pmf", _fp), ("emitter_cmf", _fp), ("emitter_pmf", _fp),
        ("emitter_sum", C.c_float),
        ("cam", _fp),
        ("sec_edge", _fp), ("sec_cmf", _fp), ("sec_pmf", _fp), ("sec_sum", C.c_float),
        ("prim_edge", _fp), ("prim_cmf", _fp), ("prim_pmf", _fp), ("prim_sum", C.c_float),
        ("guide_reso", C.c_int32 * 3),
        ("guide_cmf", _fp), ("guide_pmf", _fp), ("guide_sum", C.c_float),
        ("env_emitter", C.c_int32), ("env_tex", C.c_int32 * 3), ("env_reso", C.c_int32 * 2),
        ("env_f", _fp), ("env_cmf", _fp), ("env_pmf", _fp), ("env_sum", C.c_float),
        ("material_mask", C.c_uint32),
        ("sec_edge_faces", _fp),
        ("prim_edge_z", _fp),
    ]


class RenderOpts(C.Structure):
    _fields_ = [
        ("integrator", C.c_int32), ("bsdf_samples", C.c_int32), ("light_samples", C.c_int32),
        ("max_depth", C.c_int32), ("hide_emitters", C.c_int32), ("field", C.c_int32),
        ("spp", C.c_int32), ("sppe", C.c_int32), ("sppse", C.c_int32),
        ("spp_begin", C.c_int32), ("spp_end", C.c_int32),
        ("sppe_begin", C.c_int32), ("sppe_end", C.c_int32),
        ("sppse_begin", C.c_int32), ("sppse_end", C.c_int32),
        ("flags", C.c_int32),
        ("rng_offset", C.c_uint64 * 3),
    ]


class Tangents(C.Structure):
    _fields_ = [("d_tri_info", _fp), ("d_texels", _fp), ("d_emitter_rad", _fp), ("d_cam_to_world", _fp),
                ("d_sec_edge", _fp), ("d_prim_edge", _fp), ("d_env_f", _fp)]


class Grads(C.Structure):
    _fields_ = [("g_tri_info", _fp), ("g_texels", _fp), ("g_emitter_rad", _fp), ("g_cam_to_world", _fp),
                ("g_sec_edge", _fp), ("g_prim_edge", _fp), ("g_env_f", _fp)]


TANGENT_FIELDS = ("tri_info", "texels", "emitter_rad", "cam_to_world", "sec_edge", "prim_edge", "env_f")

# every symbol include/psdr_hip.h declares (checked by tests/test_abi.py)
HIP_SYMBOLS = (
    "psdr_last_error", "psdr_version", "psdr_abi_struct_sizes", "psdr_scene_create", "psdr_scene_destroy", "psdr_scene_set_option", "psdr_scene_set_tables",
    "psdr_bvh_build", "psdr_bvh_stats", "psdr_scene_info", "psdr_trace", "psdr_render_c", "psdr_render_d_fwd", "psdr_render_d_rev",
    "psdr_guide_build", "psdr_get_counters",
    "psdr_geo_world_vertices_fwd", "psdr_geo_world_vertices_rev", "psdr_geo_tri_rows_fwd", "psdr_geo_tri_rows_rev", "psdr_geo_sec_edges_fwd", "psdr_geo_sec_edges_rev", "psdr_geo_prim_edges_fwd", "psdr_geo_prim_edges_rev",
    "psdr_geo_compact_edges_fwd", "psdr_geo_compact_edges_rev", "psdr_geo_emitter_tables",
)

HIP_LIB_PATH = os.path.join(PKG_ROOT, "lib", "libpsdr_hip.so")   # the in-tree build; the package reads NO environment variable


def use_library(path):
    """Developer hook (tools/, tests/conftest.py: A/B runs of tools/build_variant_lib.sh builds): load THIS build of libpsdr_hip.so instead of the
    in-tree one.  Must be called before the first render call of the process."""
    global HIP_LIB_PATH
    if _hip is not None:
        raise RuntimeError("psdr_cuda._abi.use_library: the library is already loaded (%s)" % HIP_LIB_PATH)
    HIP_LIB_PATH = os.path.abspath(path)


_hip = None


def load_hip():
    """Load the product library.  Raises RuntimeError if it has not been built (no fallback)."""
    global _hip
    if _hip is not None:
        return _hip
    if not os.path.exists(HIP_LIB_PATH):
        raise RuntimeError("libpsdr_hip.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`"
                           % HIP_LIB_PATH)
    lib = C.CDLL(HIP_LIB_PATH)
    lib.psdr_last_error.restype = C.c_char_p
    lib.psdr_version.restype = C.c_char_p
    vp, i32 = C.c_void_p, C.c_int32
    lib.psdr_scene_create.argtypes = [C.POINTER(vp)]
    lib.psdr_scene_destroy.argtypes = [vp]
    lib.psdr_scene_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    lib.psdr_scene_set_tables.argtypes = [vp, C.POINTER(SceneDesc)]
    lib.psdr_bvh_build.argtypes = [vp, vp]
    lib.psdr_trace.argtypes = [vp, i32] + [vp] * 7 + [vp] * 4 + [vp]
    lib.psdr_render_c.argtypes = [vp, C.POINTER(RenderOpts), vp, vp]
    lib.psdr_render_d_fwd.argtypes = [vp, C.POINTER(RenderOpts), i32, C.POINTER(Tangents), vp, vp, vp]
    lib.psdr_render_d_rev.argtypes = [vp, C.POINTER(RenderOpts), vp, vp, C.POINTER(Grads), vp]
    lib.psdr_guide_build.argtypes = [vp, C.POINTER(RenderOpts), C.POINTER(i32), i32, vp, vp]
    lib.psdr_get_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.psdr_scene_info.argtypes = [vp, C.POINTER(i32)]
    lib.psdr_geo_world_vertices_fwd.argtypes = [i32, vp, vp, vp, vp, vp]
    lib.psdr_geo_world_vertices_rev.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp]
    lib.psdr_geo_tri_rows_fwd.argtypes = [i32, i32, vp, vp, vp, vp, i32, vp]
    lib.psdr_geo_tri_rows_rev.argtypes = [i32, i32, vp, vp, vp, vp, i32, vp, vp, vp]
    lib.psdr_geo_sec_edges_fwd.argtypes = [i32, vp, vp, vp, i32, vp, vp, vp]
    lib.psdr_geo_sec_edges_rev.argtypes = [i32, vp, vp, vp, vp, i32, vp]
    lib.psdr_geo_prim_edges_fwd.argtypes = [i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.psdr_geo_prim_edges_rev.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp]
    lib.psdr_geo_compact_edges_fwd.argtypes = [i32, vp, i32, vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.psdr_geo_compact_edges_rev.argtypes = [i32, i32, vp, vp, vp, vp]
    lib.psdr_geo_emitter_tables.argtypes = [i32, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    for name in HIP_SYMBOLS:
        if name not in ("psdr_last_error", "psdr_version"):
            getattr(lib, name).restype = C.c_int
    sizes = (C.c_int32 * 4)()
    lib.psdr_abi_struct_sizes(sizes)
    mine = (C.sizeof(SceneDesc), C.sizeof(RenderOpts), C.sizeof(Tangents), C.sizeof(Grads))
    if tuple(sizes) != mine:
        raise RuntimeError("psdr_cuda/_abi.py is out of sync with include/psdr_hip.h: %s vs %s" % (tuple(sizes), mine))
    _hip = lib
    return lib


def check(lib, rc):
    if rc != 0:
        msg = lib.psdr_last_error()
        raise RuntimeError(msg.decode() if msg else "psdr_hip call failed (rc=%d)" % rc)


def scene_stats(handle):
    """psdr_scene_info as a dict (what psdr_bvh_build chose for the handle's current tables)."""
    lib = load_hip()
    out = (C.c_int32 * 8)()
    check(lib, lib.psdr_scene_info(handle, out))
    return {"n_tiny": out[0], "n_blas": out[1], "n_inline": out[2], "leaf_tris": out[3], "device_built": out[4], "n_slab": out[5], "occ_rows": out[6], "occ_max_rows": out[7]}


def make_opts(integrator=INTEGRATOR_DIRECT, bsdf_samples=1, light_samples=1, max_depth=1, hide_emitters=False,
              field=0, spp=1, sppe=0, sppse=0, spp_range=None, sppe_range=None, sppse_range=None,
              rng_offset=(0, 0, 0), flags=0):
    o = RenderOpts()
    o.flags = flags
    o.integrator, o.bsdf_samples, o.light_samples = integrator, bsdf_samples, light_samples
    o.max_depth, o.hide_emitters, o.field = max_depth, int(bool(hide_emitters)), field
    o.spp, o.sppe, o.sppse = spp, sppe, sppse
    o.spp_begin, o.spp_end = spp_range if spp_range is not None else (0, spp)
    o.sppe_begin, o.sppe_end = sppe_range if sppe_range is not None else (0, sppe)
    o.sppse_begin, o.sppse_end = sppse_range if sppse_range is not None else (0, sppse)
    for i in range(3):
        o.rng_offset[i] = int(rng_offset[i])
    return o


def draws_per_slot(opts):
    """RNG draws consumed per slot of sampler 0/1/2 by one render call (all lanes draw, masked or
    not, exactly as Enoki does: src/integrator/direct.cpp:64-160, integrator.cpp:83,103-110)."""
    if opts.integrator == INTEGRATOR_DIRECT:
        li = 3 * opts.bsdf_samples + 2 * opts.light_samples
    elif opts.integrator == INTEGRATOR_PATH:
        li = 5 * opts.max_depth
    else:
        li = 0
    return (2 + li, 1 + 2 * li, 3)
