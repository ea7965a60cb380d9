#!/usr/bin/env python
"""Full-size gradient-projection fixtures (BASELINE configs C3 / C4): tests/golden/proj_*.npz.

What is stored: for a fixed adjoint image A (loss L = <A, image>) and N tangent fields v_i over the vertex
positions of one mesh, the numbers  b_i = <A, d image / d P_i>  with vertices = V0 + P_i v_i, evaluated by the CPU
oracle in FP64, FORWARD mode, all three terms (interior + primary-edge + secondary-edge), at the full size of the
config (bunny 512x512, spp = sppe = sppse = 128: 100 M sample slots per tangent).  The GPU test
(tests/test_projections_gpu.py) computes the same numbers the other way round -- ONE reverse-mode launch of the HIP
kernels (gradient scatter-add into the triangle / edge tables), pulled back to the vertices through the torch table
chain, then <g_V, v_i> -- and asserts a relative error <= 1e-3.  Single fp32 sample flips between two implementations
(DESIGN.md "numerical fragility") average out in these sums; per-pixel comparisons of bunny derivative images do not.

Like the rest of tests/golden these are fixtures of this build's own restatement ("parity unpinned": the reference
holds no vectors and cannot be built here).  Regenerate on a many-core host (the GPU box: ~3 s per
tangent on 256 threads; 8 cores: ~1 min per tangent):

    python tests/golden/make_projections.py c3_bunny_light c3_cbox_bunny c4_share
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("psdr-cuda_amd", "oracle", "tests"):
    if os.path.join(ROOT, p) not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, p))

GOLD = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: scene, mesh id whose vertices move, res, spp (global), sppe, sppse, spp shard, integrator kwargs, tangents
    "c3_bunny_light": dict(scene="bunny_light", mesh=0, res=512, spp=128, sppe=128, sppse=128, shard=None, ntan=32,
                           kw=dict(bsdf_samples=1, light_samples=1)),
    "c3_cbox_bunny": dict(scene="cbox_bunny", mesh=1, res=512, spp=128, sppe=128, sppse=128, shard=None, ntan=32,
                          kw=dict(bsdf_samples=1, light_samples=1)),
    # one GPU's share of C4 (1024x1024, global spp 512, rank 1 of 8 renders samples [64, 128) of every pixel)
    "c4_share": dict(scene="cbox_bunny", mesh=1, res=1024, spp=512, sppe=512, sppse=512, shard=(1, 8), ntan=8,
                     kw=dict(bsdf_samples=1, light_samples=1)),
}


def adjoint_image(res, seed=11):
    """Fixed, positive, O(1) adjoint: dL/d image."""
    g = torch.Generator().manual_seed(seed)
    return (0.5 + torch.rand((res * res, 3), generator=g)).numpy().astype(np.float32)


def tangent_fields(V0, n, seed=5):
    """n vertex tangent fields [Nv, 3] (float32, CPU): rigid motions first (translations, rotations about the
    centroid, a uniform scale), then smooth seeded sinusoidal displacement fields, the last quarter white noise."""
    V = V0.detach().cpu().double()
    c = V.mean(0)
    ext = float((V.max(0).values - V.min(0).values).max())
    X = (V - c) / ext
    g = torch.Generator().manual_seed(seed)
    out = []
    eye = torch.eye(3, dtype=torch.float64)
    for k in range(3):
        out.append(eye[k].expand_as(V).clone())
    for k in range(3):
        out.append(torch.cross(eye[k].expand_as(V), X, dim=1))
    out.append(X.clone())
    while len(out) < n - n // 4:
        freq = torch.randint(1, 4, (3,), generator=g).double() * 3.14159265
        phase = torch.rand(3, generator=g, dtype=torch.float64) * 6.2831853
        amp = torch.randn(3, generator=g, dtype=torch.float64)
        s = torch.sin((X * freq + phase).sum(1, keepdim=True))
        out.append(s * amp)
    while len(out) < n:
        out.append(torch.randn(V.shape, generator=g, dtype=torch.float64))
    return [t.float() for t in out[:n]]


def build_scene(case, device, vertex_param=None):
    """Scene of `case` with the moving mesh's raw vertex positions = vertex_param(V0) (a torch tensor that may carry
    a graph); returns (scene, tables, V0)."""
    import psdr_cuda
    from enoki.cuda_autodiff import Vector3f as Vector3fD
    from psdr_cuda.fixtures import scene_path
    c = CASES[case]
    sc = psdr_cuda.Scene()
    sc.load_file(scene_path(c["scene"]), False)
    sc.opts.width = sc.opts.height = c["res"]
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse, sc.opts.log_level = c["spp"], c["sppe"], c["sppse"], 0
    m = sc.m_meshes[c["mesh"]]
    V0 = m._vertex_positions_raw.detach().clone()
    if vertex_param is not None:
        m.vertex_positions = Vector3fD._wrap(vertex_param(V0))
    sc.configure()
    return sc, sc.tables(0), V0


def render_opts(case):
    from psdr_cuda import _abi
    from psdr_cuda.integrator import shard_range
    c = CASES[case]
    rng = {}
    if c["shard"] is not None:
        r, w = c["shard"]
        rng = dict(spp_range=shard_range(c["spp"], r, w), sppe_range=shard_range(c["sppe"], r, w), sppse_range=shard_range(c["sppse"], r, w))
    return _abi.make_opts(spp=c["spp"], sppe=c["sppe"], sppse=c["sppse"], **rng, **c["kw"])


def oracle_projections(case, ntan=None, precision=1, verbose=True, reference_form=True):
    """fp64 + the reference's literal forms by default: the exact value of the reference's estimator."""
    import oracle
    from enoki._array import _jvp_wrt
    from helpers import AD_KEYS
    c = CASES[case]
    ntan = ntan or c["ntan"]
    adj = adjoint_image(c["res"]).astype(np.float64)
    o = render_opts(case)
    out, scale = [], []
    fields = None
    for i in range(ntan):
        P = torch.zeros((), dtype=torch.float32, requires_grad=True)
        holder = {}

        def param(V0):
            holder["v"] = tangent_fields(V0, ntan)[i].to(V0.device)
            return V0 + holder["v"] * P
        sc, tb, V0 = build_scene(case, "cpu", param)
        n_edges = int(tb["num_sec_edges"])
        tan = dict(zip(AD_KEYS, _jvp_wrt([tb.get(k) for k in AD_KEYS], P)))
        t0 = time.time()
        img, dimg = oracle.render(tb, o, mode=1, tangents=tan, precision=precision, reference_form=reference_form)
        d = dimg.astype(np.float64)
        out.append(float((adj * d).sum()))
        scale.append(float(np.abs(adj * d).sum()))
        if verbose:
            print("%s tangent %2d: <A, dI> = %+.9e   sum|A dI| = %.3e   (%.1f s)" % (case, i, out[-1], scale[-1], time.time() - t0), flush=True)
    return np.array(out), np.array(scale), float(img.astype(np.float64).mean()), n_edges


def main():
    for case in sys.argv[1:] or list(CASES):
        b, s, mean, n_edges = oracle_projections(case)
        c = CASES[case]
        # num_sec_edges: the secondary-edge list the sample streams follow (the native table chain decides every borderline coplanar edge
        # deterministically; the test asserts that its tables hold the same list)
        np.savez(os.path.join(GOLD, "proj_%s.npz" % case), b=b, scale=s, image_mean=mean, ntan=len(b), res=c["res"], spp=c["spp"],
                 sppe=c["sppe"], sppse=c["sppse"], num_sec_edges=n_edges)
        print("wrote proj_%s.npz" % case)


if __name__ == "__main__":
    main()
