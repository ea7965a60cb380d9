"""EXR reader / writer (replaces tinyexr behind Bitmap::load_openexr, reference src/core/bitmap_loader.cpp)."""
import os

import numpy as np

import psdr_cuda
from psdr_cuda import _abi
from psdr_cuda.exr import load_exr_rgba, save_exr_rgb
from psdr_cuda.fixtures import DATA_DIR, scene_path


def test_roundtrip_zip_and_raw(tmp_path):
    img = np.random.default_rng(0).random((37, 53, 3)).astype(np.float32) * 5 - 1
    for comp in (True, False):
        p = str(tmp_path / ("a%d.exr" % comp))
        save_exr_rgb(p, img, comp)
        out, (w, h) = load_exr_rgba(p)
        assert (w, h) == (53, 37) and np.array_equal(out[..., :3], img) and np.all(out[..., 3] == 1.0)


def test_reference_texture_known_values():
    """Known-answer values of the reference's test asset (decoded here with zlib; the reference's own
    tests hold the file but no expected numbers)."""
    a, (w, h) = load_exr_rgba(os.path.join(DATA_DIR, "textures", "test_texture.exr"))
    assert (w, h) == (512, 512) and a.dtype == np.float32
    assert np.all(a[..., 2] == 0.0) and np.all(a[..., 3] == 1.0)
    assert abs(float(a[..., 0].mean()) - 0.16082188) < 1e-4 and abs(float(a[..., 1].mean()) - 0.16082466) < 1e-4     # float32 mean: summation-order dependent
    assert a.min() >= 0.0 and a.max() <= 1.0 + 1e-6


def test_bitmap_and_xml_texture_loading():
    b = psdr_cuda.Bitmap3fD(os.path.join(DATA_DIR, "textures", "test_texture.exr"))
    assert b.resolution == (512, 512) and b.data.numpy().shape == (512 * 512, 3)
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path("cbox_uv_exr"), False)
    sc.opts.width = sc.opts.height = 16
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = 2, 0, 0, 0
    sc.configure()
    tb = sc.tables(0)
    rec = tb["bsdf_rec"].cpu().numpy()
    assert any(r[2] == 512 and r[3] == 512 for r in rec)
    assert tb["texels"].numel() >= 512 * 512 * 3
    import oracle
    img = oracle.render(tb, _abi.make_opts(spp=2))
    assert np.isfinite(img).all() and img.mean() > 0.05


def test_cv2_stand_in(tmp_path):
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("cv2_shim", os.path.join(_abi.PKG_ROOT, "cv2", "__init__.py"))
    cv2 = importlib.util.module_from_spec(spec); spec.loader.exec_module(cv2)
    img = np.random.default_rng(1).random((8, 9, 3)).astype(np.float32)
    p = str(tmp_path / "o.exr")
    assert cv2.imwrite(p, cv2.cvtColor(img, cv2.COLOR_RGB2BGR))
    out, _ = load_exr_rgba(p)
    assert np.array_equal(out[..., :3], img)       # RGB on disk
