import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "psdr-cuda_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once (hipcc cross-compiles
    # without a GPU; on the GPU box the prebuilt files travel with the snapshot)
    libs = (os.path.join(ROOT, "psdr-cuda_amd", "lib", "libpsdr_hip.so"), os.path.join(ROOT, "oracle", "libpsdr_oracle.so"),
            os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so"))
    if not all(os.path.exists(p) for p in libs):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(autouse=True)
def _poison_scratch(request):
    """PSDR_TEST_POISON=<hex pattern> (developer switch): before every GPU test the scratch arena of the queue is filled with the pattern
    (tests/poison/poison.hip) -- a kernel that reloads a spill slot it never stored for some lane then computes with the pattern instead of
    with whatever an earlier kernel left there (the order-dependent gradient).  tests/test_rough_rev_order_gpu.py does this
    for the kernels the defect was found in, whatever the environment says."""
    pat = os.environ.get("PSDR_TEST_POISON")
    if pat and "gpu" in request.keywords:
        from helpers import poison_gpu
        poison_gpu(int(pat, 16), 1)
    yield


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
