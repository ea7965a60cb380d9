"""round 6: bunny_env_2 of the reference's scenario table, AD against central differences term by term (interior / primary edges / secondary edges)."""
import sys, os, numpy as np, tempfile, pathlib
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "psdr-cuda_amd")); sys.path.insert(0, R)
import ref_harness as H
name = sys.argv[1] if len(sys.argv) > 1 else "bunny_env_2"
args = H.SCENARIOS[name]; fdc = args["FD"]
tmp = pathlib.Path(tempfile.mkdtemp())
np.set_printoptions(linewidth=250, precision=0, suppress=True)
def blocks(a, Hh, W, B=54):
    bh, bw = Hh // B, W // (B * 480 // 270 if False else B)
    return a[:bh * B, :bw * B].reshape(bh, B, bw, B, 3).sum(axis=(1, 3, 4))
sc = H.load(args, tmp); W, Hh = sc.opts.width, sc.opts.height
fd = H.run_fd(H.make_integrator(args), args, tmp, 64).reshape(Hh, W, 3)
print("FD sum %.4g" % fd.sum()); print(blocks(fd, Hh, W))
tot = 0
for label, cnt in (("interior", (16, 0, 0)), ("primary", (0, 16, 0)), ("secondary", (0, 0, 64)), ("all", (4, 4, 64))):
    ad = dict(args["AD"]); ad["spp"], ad["sppe"], ad["sppse"] = cnt
    integ = H.make_integrator(args); sc = H.load(args, tmp)
    d = H.run_ad(integ, sc, ad, 32).reshape(Hh, W, 3)
    print("AD %-9s spp/sppe/sppse %s: sum %.4g" % (label, cnt, d.sum())); print(blocks(d, Hh, W))
