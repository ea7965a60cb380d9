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


def main():
    for name in CASES:
        sc, inp, img, dimg, prim, tang = run_case(name)
        np.savez(os.path.join(GOLD, name + ".npz"), img=img, dimg=dimg, tri_info=prim["tri_info"].numpy(), cam=prim["cam"].numpy(),
                 d_tri_info=tang["tri_info"].numpy())
        print(name, img.shape, float(np.abs(img).mean()), float(np.abs(dimg).max()))


if __name__ == "__main__":
    main()
