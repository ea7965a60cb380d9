#!/usr/bin/env python
"""tests/golden/torch_*.npz: image and derivative image of the SECOND oracle (oracle/torch_oracle.py: torch, fp64,
its own table chain from the raw scene inputs, brute-force hits, forward-mode AD) for small renderD problems --
DirectIntegrator(1, 1), all three terms, one mesh translated along a direction.  The C++ oracle and the GPU are both
checked against these files (tests/test_second_oracle.py), so a fixture does not depend on the implementation it tests.

    python tests/golden/make_torch_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    if os.path.join(ROOT, p) not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, p))
GOLD = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: scene, res, spp, sppe, sppse, moving mesh, direction
    "torch_cbox_d": ("cbox", 16, 4, 4, 4, 0, (1.0, 0.5, 0.0)),
    "torch_cbox_occluder_d": ("cbox_occluder", 16, 4, 4, 4, 1, (1.0, 0.5, 0.0)),
}


def run_case(name):
    import torch_oracle as to
    from helpers import load_scene
    scene, res, spp, sppe, sppse, mesh, direction = CASES[name]
    sc, _ = load_scene(scene, res=res, spp=spp, sppe=sppe, sppse=sppse)
    inp = to.scene_inputs(sc)
    img, dimg, prim, tang = to.render_d(inp, mesh, direction, spp=spp, sppe=sppe, sppse=sppse)
    return sc, inp, img.numpy(), dimg.numpy(), prim, tang


# rough conductors / the PathTracer (round 6): the torch oracle's table chain builds diffuse scenes only, so these fixtures are ITS renderer on the PRODUCT's fp32 tables
# (whose chain the diffuse fixtures and test_product_tables_equal_the_independent_tables check): image and derivative image w.r.t. every texel (a seeded tangent)
TABLE_CASES = {
    # name: scene, res, spp, renderer arguments of torch_oracle.render, the same as C-ABI options
    "torch_cbox_rough_path3": ("cbox_rough", 24, 8, dict(depth=3), dict(integrator=1, max_depth=3)),          # 1 = PSDR_INTEGRATOR_PATH,
    "torch_cbox_rough_direct22": ("cbox_rough", 24, 8, dict(B=2, L=2), dict(bsdf_samples=2, light_samples=2)),
}


def texel_tangent(n):
    import torch
    return torch.rand(n, generator=torch.Generator().manual_seed(17)) - 0.3


def run_table_case(name):
    import torch
    import torch_oracle as to
    from helpers import load_scene
    scene, res, spp, kw_t, _ = TABLE_CASES[name]
    sc, _ = load_scene(scene, res=res, spp=spp)
    tb = sc.tables(0)
    tbc = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in tb.items()}
    img, dimg = to.render_d_from_tables(tbc, {"texels": texel_tangent(tbc["texels"].numel())}, spp=spp, **kw_t)
    return tb, img.numpy(), dimg.numpy()


def main():
    for name in TABLE_CASES:
        tb, img, dimg = run_table_case(name)
        np.savez(os.path.join(GOLD, name + ".npz"), img=img, dimg=dimg, texels=tb["texels"].detach().cpu().numpy())
        print(name, img.shape, float(np.abs(img).mean()), float(np.abs(dimg).max()))
    for name in CASES:
        sc, inp, img, dimg, prim, tang = run_case(name)
        np.savez(os.path.join(GOLD, name + ".npz"), img=img, dimg=dimg, tri_info=prim["tri_info"].numpy(), cam=prim["cam"].numpy(),
                 d_tri_info=tang["tri_info"].numpy())
        print(name, img.shape, float(np.abs(img).mean()), float(np.abs(dimg).max()))


if __name__ == "__main__":
    main()
