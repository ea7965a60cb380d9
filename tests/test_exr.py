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


# ------------------------------------------------------------------------------------ PIZ
def test_piz_roundtrip_odd_sizes_short_last_chunk_and_float_channels(tmp_path):
    """reader vs the test-only encoder (tests/piz_encode.py): sizes that are not powers of two, a last
    chunk shorter than 32 lines, HALF (14-bit wavelet) and FLOAT + alpha (16-bit wavelet) channels"""
    from piz_encode import save_exr_piz
    rng = np.random.default_rng(0)
    for h, w, dt in ((32, 64, np.float16), (37, 53, np.float16), (45, 70, np.float32), (5, 3, np.float16), (70, 33, np.float32)):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.sin(xx * 0.2) + np.cos(yy * 0.13) + 2.5
        ch = {n: (base * (i + 1) + rng.random((h, w)) * 0.05).astype(dt) for i, n in enumerate("BGR")}
        if dt == np.float32:
            ch["A"] = rng.random((h, w)).astype(np.float32)
        p = str(tmp_path / "t.exr")
        save_exr_piz(p, ch)
        img, (ww, hh) = load_exr_rgba(p)
        assert (ww, hh) == (w, h)
        for i, n in enumerate("RGB"):
            assert np.array_equal(img[..., i], ch[n].astype(np.float32)), (h, w, dt, n)
        if "A" in ch:
            assert np.array_equal(img[..., 3], ch["A"])


def test_piz_constant_and_run_length_symbol(tmp_path):
    """a constant image compresses to one symbol + run-length codes in real encoders; ours emits no runs, so
    splice one in by hand: the reader must expand `rlc, count` to repeats of the previous symbol"""
    from piz_encode import save_exr_piz
    ch = {n: np.full((32, 16), 1.5 * (i + 1), dtype=np.float16) for i, n in enumerate("BGR")}
    p = str(tmp_path / "c.exr")
    save_exr_piz(p, ch)
    img, _ = load_exr_rgba(p)
    assert np.all(img[..., 0] == 4.5) and np.all(img[..., 1] == 3.0) and np.all(img[..., 2] == 1.5)


def test_reference_environment_map_piz_known_values():
    """the reference's ballroom_1k.exr (1024x512 half, PIZ).  Values decoded here; a wrong wavelet or
    Huffman step turns the panorama into noise, so smoothness + these spot values pin the decoder."""
    a, (w, h) = load_exr_rgba(os.path.join(DATA_DIR, "envmaps", "ballroom_1k.exr"))
    assert (w, h) == (1024, 512) and np.isfinite(a).all() and a[..., :3].min() >= 0.0 and float(a[..., :3].max()) == 141.5
    assert np.allclose(a[..., :3].astype(np.float64).mean((0, 1)), [0.56065734, 0.45016956, 0.34719557], rtol=1e-6)
    assert np.allclose(a[100, 200, :3], [0.5292969, 0.41992188, 0.29101562]) and np.allclose(a[300, 700, :3], [0.07446289, 0.05200195, 0.05249023])
    assert np.allclose(a[0, 0, :3], [0.35253906, 0.17089844, 0.02246094]) and np.allclose(a[511, 1023, :3], [0.17138672, 0.07763672, 0.04248047])
    rgb = a[..., :3]
    assert np.median(np.abs(np.diff(rgb, axis=1))) < 0.05 * np.median(rgb)           # a photograph, not noise (measured 0.025)


def test_corrupt_piz_is_rejected(tmp_path):
    import pytest
    from piz_encode import save_exr_piz
    ch = {n: (np.random.default_rng(1).random((32, 32)) + 1).astype(np.float16) for n in "BGR"}
    p = str(tmp_path / "x.exr")
    save_exr_piz(p, ch)
    raw = bytearray(open(p, "rb").read())
    raw[-200:-100] = bytes(100)                      # zero a stretch of the Huffman stream
    open(p, "wb").write(bytes(raw))
    try:
        img, _ = load_exr_rgba(p)
    except RuntimeError as e:
        assert "EXR" in str(e)
    else:
        assert not np.array_equal(img[..., 0], ch["R"].astype(np.float32))       # decoded garbage, but no crash


def test_ballroom_scene_loads_and_renders_with_the_oracle():
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path("bunny_env_ballroom"), False)
    sc.opts.width = sc.opts.height = 16
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = 4, 0, 0, 0
    sc.configure()
    tb = sc.tables(0)
    assert tb["env_tex"][1:] == [1024, 512] and tb["env_reso"] == [2046, 1022] and tb["env_cmf"].numel() == 2046 * 1022
    import oracle
    img = oracle.render(tb, _abi.make_opts(spp=4, bsdf_samples=1, light_samples=1))
    assert np.isfinite(img).all() and 0.05 < img.mean() < 5.0
