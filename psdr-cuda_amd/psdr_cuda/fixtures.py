"""Locations of the bundled scene fixtures (psdr-cuda_amd/data, see its README)."""
import os

from ._abi import PKG_ROOT

DATA_DIR = os.path.join(PKG_ROOT, "data")


def scene_path(name):
    p = os.path.join(DATA_DIR, "scenes", name if name.endswith(".xml") else name + ".xml")
    if not os.path.exists(p):
        raise RuntimeError("unknown scene fixture: " + name)
    return p
