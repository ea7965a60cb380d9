#!/usr/bin/env python
"""Developer tool (GPU box): where the GPU-vs-fp64-oracle difference of the vertex-gradient projections comes from --
per term (interior / primary edge / secondary edge): the oracle in fp32 in the reference's LITERAL forms, the oracle in fp32 in
the product's robust forms, GPU forward and reverse mode -- all against the oracle in fp64, literal forms (the exact
value of the reference's estimator).  Columns are errors relative to sum|A dI| of the term."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("psdr-cuda_amd", "oracle", "tests", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import oracle
import make_projections as mp
from enoki._array import _jvp_wrt
from helpers import GpuScene, AD_KEYS
from psdr_cuda import _abi

case = sys.argv[1] if len(sys.argv) > 1 else "c3_bunny_light"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ntan = int(sys.argv[4]) if len(sys.argv) > 4 else 6
which = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else list(range(ntan))
mp.CASES[case].update(res=res, spp=spp, sppe=spp, sppse=spp, shard=None)
adj = mp.adjoint_image(res)
holder = {}
def param(V0):
    holder["V"] = V0.clone().requires_grad_(True); return holder["V"]
sc, tb, V0 = mp.build_scene(case, "cuda", param)
g = GpuScene(tb)
fields = mp.tangent_fields(V0, ntan)
for name, (a_, b_, c_) in (("interior", (spp, 0, 0)), ("primary", (0, spp, 0)), ("secondary", (0, 0, spp)), ("all", (spp, spp, spp))):
    o = _abi.make_opts(spp=a_, sppe=b_, sppse=c_, bsdf_samples=1, light_samples=1)
    _, grads = g.render_d_rev(o, adj, want=["tri_info", "sec_edge", "prim_edge"], with_image=False)
    outs = [tb[k] for k in ("tri_info", "sec_edge", "prim_edge")]
    gV = torch.autograd.grad(outs, holder["V"], grad_outputs=[torch.as_tensor(grads[k], device="cuda") for k in ("tri_info", "sec_edge", "prim_edge")], retain_graph=True)[0].double().cpu()
    rows = []
    for i in which:
        P = torch.zeros((), requires_grad=True)
        sc2, tb2, _ = mp.build_scene(case, "cpu", lambda V: V + fields[i].to(V.device) * P.to(V.device))
        tan = dict(zip(AD_KEYS, _jvp_wrt([tb2.get(k) for k in AD_KEYS], P)))
        d64 = oracle.render(tb2, o, mode=1, tangents=tan, precision=1)[1].astype(np.float64)
        d32 = oracle.render(tb2, o, mode=1, tangents=tan, precision=0)[1].astype(np.float64)
        d32lit = oracle.render(tb2, o, mode=1, tangents=tan, precision=0, reference_form=True)[1].astype(np.float64)
        d64 = oracle.render(tb2, o, mode=1, tangents=tan, precision=1, reference_form=True)[1].astype(np.float64)
        dg = g.render_d_fwd(o, [tan])[1][0].astype(np.float64)
        A = adj.astype(np.float64)
        b64, b32, af, b32lit = (A * d64).sum(), (A * d32).sum(), (A * dg).sum(), (A * d32lit).sum()
        ar = float((gV * fields[i].double()).sum())
        sc_ = np.abs(A * d64).sum()
        rows.append((b64, (b32lit - b64) / sc_, (b32 - b64) / sc_, (af - b64) / sc_, (ar - b64) / sc_, (ar - af) / sc_, sc_))
    print("== %s  (errors relative to sum|A dI|)" % name)
    print("      b_fp64(lit)   fp32lit-64    fp32robust-64 gpu_fwd-64    gpu_rev-64    gpu_rev-fwd   scale")
    for r in rows:
        print("  %+.6e  %+.2e  %+.2e  %+.2e  %+.2e  %+.2e  %.3e" % r)
