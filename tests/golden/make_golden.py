#!/usr/bin/env python
"""Generates tests/golden/*.npz: small input/output vectors of the CPU oracle (fp32 arithmetic, the
reference's type).  The reference itself ships no golden vectors and cannot run here,
so these are REGRESSION fixtures of this build's own restatement ("parity unpinned"): they freeze the
oracle's behaviour so that later edits to oracle/ or to the table builder cannot drift unnoticed, and
they give the GPU tests a fixed target that does not need the oracle to be rebuilt.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import oracle  # noqa: E402
from helpers import load_scene, tangents_wrt  # noqa: E402
from psdr_cuda import _abi  # noqa: E402

CASES = {
    # name: (scene, res, spp, sppe, sppse, translate, opts kwargs)
    "cbox_direct11_c": ("cbox", 32, 4, 0, 0, None, dict(bsdf_samples=1, light_samples=1)),
    "cbox_rough_direct22_c": ("cbox_rough", 32, 2, 0, 0, None, dict(bsdf_samples=2, light_samples=2)),
    "cbox_path3_c": ("cbox", 32, 4, 0, 0, None, dict(integrator=_abi.INTEGRATOR_PATH, max_depth=3)),
    "cbox_occluder_direct11_d": ("cbox_occluder", 24, 4, 4, 4, (1, (1.0, 0.5, 0.0)), dict(bsdf_samples=1, light_samples=1)),
    "cbox_field_depth_c": ("cbox", 32, 1, 0, 0, None, dict(integrator=_abi.INTEGRATOR_FIELD, field=_abi.FIELDS["depth"])),
    # environment map (translate = "env_rotate": the parameter is the map's rotation angle about y)
    "bunny_env_direct11_c": ("bunny_env", 24, 4, 0, 0, None, dict(bsdf_samples=1, light_samples=1)),
    "bunny_env_rotate_d": ("bunny_env", 24, 4, 0, 0, "env_rotate", dict(bsdf_samples=1, light_samples=1)),
}


def run_case(name):
    scene, res, spp, sppe, sppse, tr, kw = CASES[name]
    if tr == "env_rotate":
        import enoki as ek
        from helpers import FloatD, Matrix4fD, Vector3fD
        sc, _ = load_scene(scene, res=res, spp=spp, sppe=sppe, sppse=sppse)
        P = FloatD(0.)
        ek.set_requires_gradient(P)
        sc.param_map["Emitter[0]"].set_transform(Matrix4fD.rotate(Vector3fD([0., 1., 0.]), P))
        sc.configure()
    else:
        sc, P = load_scene(scene, res=res, spp=spp, sppe=sppe, sppse=sppse, translate=tr)
    tb = sc.tables(0)
    o = _abi.make_opts(spp=spp, sppe=sppe, sppse=sppse, **kw)
    if tr is None:
        return tb, o, None, oracle.render(tb, o), None
    tan = tangents_wrt(tb, P)
    img, dimg = oracle.render(tb, o, mode=1, tangents=tan)
    return tb, o, tan, img, dimg


def table_digest(tb):
    """what pins the table builder: the triangle table itself when small, else its column sums; the
    environment-map record and cell masses when there is one"""
    t = tb["tri_info"].detach().numpy()
    d = {"tri_info": t} if t.shape[0] <= 64 else {"tri_info_colsum": t.astype(np.float64).sum(0), "num_tris": np.int64(t.shape[0])}
    if tb.get("env_emitter", -1) >= 0:
        d["env_f"] = tb["env_f"].detach().numpy()
        d["env_pmf_rowsum"] = tb["env_pmf"].numpy().reshape(tb["env_reso"][0], tb["env_reso"][1]).astype(np.float64).sum(0)
    return d


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        tb, o, tan, img, dimg = run_case(name)
        data = {"img": img, "cam": tb["cam"].numpy()}
        data.update(table_digest(tb))
        if dimg is not None:
            data["dimg"] = dimg
            for k, v in tan.items():
                if v is not None:
                    data["d_" + k] = v.numpy()
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **data)
        print(name, img.shape, float(img.mean()))
    data = {"rng_slot7_off3": oracle.rng(7, 3, 16), "rng_slot0": oracle.rng(0, 0, 16), "rng_slot_big": oracle.rng(2 ** 31 + 5, 11, 16)}
    np.savez_compressed(os.path.join(out_dir, "rng_streams.npz"), **data)


if __name__ == "__main__":
    main()
