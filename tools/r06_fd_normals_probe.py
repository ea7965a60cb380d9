"""round 6: AD against central differences on the bunny's own pixels (reference file cbox_bunny.xml at 128^2, the bunny translated along x / rotated about z),
with the file's SMOOTH normals and with face normals on the bunny.  With interpolated normals on a coarse mesh the lighting cut-off of a surface point follows
its FACE's horizon, so the shading jumps across every mesh edge; perspective.cpp:57-66 puts those edges into the primary-edge table only for face-normal
meshes -- for smooth meshes the estimator (the reference's, restated here) carries no boundary term for them."""
import sys, os, numpy as np, tempfile, pathlib
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "psdr-cuda_amd")); sys.path.insert(0, R)
import ref_harness as H
tmp = pathlib.Path(tempfile.mkdtemp())
np.set_printoptions(linewidth=250, precision=2, suppress=True)
def blocks(a, Hh, W, n=8):
    B = Hh // n
    return a[:n * B, :n * B].reshape(n, B, n, B, 3).sum(axis=(1, 3, 4))[4:8, 2:6]
def load(args, face):
    sc = H.load(args, tmp, res=128)
    sc.param_map["Mesh[1]"].use_face_normals = face
    return sc
def fd(args, face, npass):
    ad, eps = args["AD"], args["FD"]["eps"]
    from enoki.cuda_autodiff import Float32 as FloatD
    integ = H.make_integrator(args); scs = []
    for sgn in (-1.0, 1.0):
        s = load(args, face); s.opts.sppe, s.opts.sppse = 0, 0
        H.apply_parameter(s, ad, FloatD(sgn * eps), {}); s.configure(); scs.append(s)
    acc = 0
    for _ in range(npass):
        acc = acc + (integ.renderC(scs[1]).numpy().astype(np.float64) - integ.renderC(scs[0]).numpy().astype(np.float64))
    return acc / (2 * eps * npass)
for label, ad0, eps in (("translate x", dict(type="mesh_transform", Mesh_ID=[1], Mesh_dir=[[1., 0., 0.]]), 0.2), ("rotate z", dict(type="mesh_rotate", Mesh_ID=[1], axis=[[0., 0., 1.]]), 0.004)):
    args = dict(test_type="direct", scene_file="cbox_bunny.xml", bsdf_samples=1, light_samples=1, AD=dict(ad0, spp=16, sppe=16, sppse=64), FD=dict(npass=64, eps=eps))
    for face in (False, True):
        sc = load(args, face); W, Hh = sc.opts.width, sc.opts.height
        f = fd(args, face, 300).reshape(Hh, W, 3)
        d = H.run_ad(H.make_integrator(args), load(args, face), args["AD"], 150).reshape(Hh, W, 3)
        bf, bd = blocks(f, Hh, W), blocks(d, Hh, W)
        print("== %s, bunny with %s normals: FD / AD block sums, |AD - FD| / |FD| over the blocks %.3f" % (label, "FACE" if face else "smooth", np.linalg.norm(bd - bf) / np.linalg.norm(bf)))
        print(bf); print(bd)
