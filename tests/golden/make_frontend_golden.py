"""Golden vectors for the hand-off rows (f2 second half, f3 decoders), produced by running the
REFERENCE's own code in this container (it cannot travel to the GPU box):

  * models/psmnet/psmnet.py:150-197  -- PSMNet.forward up to the tensor handed to dres0 (concat volume
    + hint modulation); feature_extraction is replaced by the identity and dres0 by a probe.
  * models/raft_stereo/corr.py:151-180 -- CorrBlock1D.corr with and without hints.
  * dataloaders/frame_utils.py:34-69 -- readPFM on a generated file, readDispKITTI with cv2.imread
    replaced by a loader of the same uint16 array (OpenCV is not installed).

Run:  python tests/golden/make_frontend_golden.py     (writes tests/golden/frontend_cases.npz)
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
torch.Tensor.cuda = lambda self, *a, **k: self          # the reference calls .cuda(); there is no GPU here

out = {}
rng = np.random.default_rng(77)


def hints_pair(B, H, W, dmax, p):
    valid = (rng.random((B, 1, H, W)) < p).astype(np.float32)
    hints = (rng.uniform(0.5, dmax - 1, (B, 1, H, W)).astype(np.float32)) * valid
    return hints, valid


# ---- PSMNet ---------------------------------------------------------------------------------
import models.psmnet.psmnet as P  # noqa: E402


class Probe(Exception):
    pass


def psm_case(tag, B, C, H, W, maxdisp, with_hints, p=0.3):
    H4, W4 = H // 4, W // 4
    net = P.PSMNet.__new__(P.PSMNet)
    torch.nn.Module.__init__(net)
    net.maxdisp = maxdisp
    net.feature_extraction = lambda x: x

    def probe(cost):
        raise Probe(cost)
    net.dres0 = probe
    fl = rng.standard_normal((B, C, H4, W4)).astype(np.float32)
    fr = rng.standard_normal((B, C, H4, W4)).astype(np.float32)
    hints, valid = hints_pair(B, H, W, maxdisp, p)
    try:
        if with_hints:
            P.PSMNet.forward(net, torch.from_numpy(fl), torch.from_numpy(fr), torch.from_numpy(hints), torch.from_numpy(valid))
        else:
            P.PSMNet.forward(net, torch.from_numpy(fl), torch.from_numpy(fr))
        raise RuntimeError("probe not reached")
    except Probe as e:
        cost = e.args[0].detach().numpy()
    out[f"psm_{tag}_fl"], out[f"psm_{tag}_fr"] = fl, fr
    out[f"psm_{tag}_hints"], out[f"psm_{tag}_valid"] = hints, valid
    out[f"psm_{tag}_meta"] = np.array([maxdisp, int(with_hints)], np.int64)
    out[f"psm_{tag}_cost"] = cost.astype(np.float32)


psm_case("a", 2, 3, 20, 36, 16, True)
psm_case("b", 1, 4, 23, 38, 24, True, p=0.6)     # H, W not multiples of 4: nearest subsampling with a fractional scale
psm_case("c", 1, 2, 16, 32, 48, True, p=1.0)     # D4 = 12 > W4 = 8: disparity planes wider than the row
psm_case("d", 2, 3, 20, 36, 16, False)

# ---- RAFT-Stereo correlation ----------------------------------------------------------------
_oe = types.ModuleType("opt_einsum")          # imported by update.py (package __init__), not used by corr.py
_oe.contract = torch.einsum
sys.modules.setdefault("opt_einsum", _oe)
import models.raft_stereo.corr as RC  # noqa: E402


def raft_case(tag, B, Dc, H, W, p=0.3):
    H4, W4 = H // 4, W // 4
    f2 = rng.standard_normal((B, Dc, H4, W4)).astype(np.float32)
    f3 = rng.standard_normal((B, Dc, H4, W4)).astype(np.float32)
    hints, valid = hints_pair(B, H, W, W4 * 4, p)
    pre = RC.CorrBlock1D.corr(torch.from_numpy(f2), torch.from_numpy(f3)).numpy()
    post = RC.CorrBlock1D.corr(torch.from_numpy(f2), torch.from_numpy(f3), torch.from_numpy(hints), torch.from_numpy(valid)).numpy()
    out[f"raft_{tag}_f2"], out[f"raft_{tag}_f3"] = f2, f3
    out[f"raft_{tag}_hints"], out[f"raft_{tag}_valid"] = hints, valid
    out[f"raft_{tag}_pre"], out[f"raft_{tag}_post"] = pre.astype(np.float32), post.astype(np.float32)


raft_case("a", 2, 8, 20, 36)
raft_case("b", 1, 16, 23, 38, p=0.7)

# ---- decoders --------------------------------------------------------------------------------
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
import cv2  # noqa: E402
_png = {}
cv2.IMREAD_ANYDEPTH = 2
cv2.imread = lambda fn, flag=None: _png[fn]
cv2.setNumThreads = lambda n: None
cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)
for name in ("imageio", "OpenEXR", "Imath", "PIL", "PIL.Image", "imageio.v2"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["PIL"].Image = sys.modules["PIL.Image"]
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("ref_frame_utils", os.path.join(REF, "dataloaders", "frame_utils.py"))
FU = importlib.util.module_from_spec(spec)
spec.loader.exec_module(FU)

u16 = rng.integers(0, 65536, (9, 14), dtype=np.uint16)
u16[rng.random(u16.shape) < 0.4] = 0
_png["k.png"] = u16
disp, valid = FU.readDispKITTI("k.png")
out["kitti_u16"], out["kitti_disp"], out["kitti_valid"] = u16, disp[..., 0].astype(np.float32), valid[..., 0]

for tag, chans, little in (("g", 1, True), ("c", 3, False)):
    Hh, Ww = 7, 11
    data = rng.standard_normal((Hh, Ww, 3) if chans == 3 else (Hh, Ww)).astype(np.float32)
    raw = data.astype("<f4" if little else ">f4").tobytes()
    with tempfile.NamedTemporaryFile(suffix=".pfm", delete=False) as f:
        f.write(b"PF\n" if chans == 3 else b"Pf\n")
        f.write(f"{Ww} {Hh}\n".encode())
        f.write(b"-1.0\n" if little else b"1.0\n")
        f.write(raw)
        fn = f.name
    dec = FU.readPFM(fn)
    os.unlink(fn)
    out[f"pfm_{tag}_raw"] = np.frombuffer(raw, np.uint8).copy()
    out[f"pfm_{tag}_meta"] = np.array([Hh, Ww, chans, int(little)], np.int64)
    out[f"pfm_{tag}_dec"] = np.ascontiguousarray(dec).astype(np.float32)

# ---- harness rows a20: losses.py:5-24 ----------------------------------------------------------
spec = importlib.util.spec_from_file_location("ref_losses", os.path.join(REF, "losses.py"))
RL = importlib.util.module_from_spec(spec)
spec.loader.exec_module(RL)
lh = rng.uniform(0, 190, (2, 1, 13, 17)).astype(np.float32)
lv = (rng.random((2, 1, 13, 17)) < 0.7).astype(np.float32)
lh[0, 0, 0, 0], lv[0, 0, 0, 0] = np.inf, 1.0       # dropped or not, never NaN
torch.manual_seed(123)
nh, nv = RL.sample_hints(torch.from_numpy(lh.copy()), torch.from_numpy(lv.copy()), 0.4)
out["loss_hints"], out["loss_valid"] = lh, lv
out["loss_new_hints"], out["loss_new_valid"] = nh.numpy(), nv.numpy()
md, mg = rng.uniform(0, 60, (19, 23)).astype(np.float32), rng.uniform(0, 60, (19, 23)).astype(np.float32)
mv = (rng.random((19, 23)) < 0.6).astype(np.float32)
m = RL.guided_metrics(md.copy(), mg.copy(), mv.copy())
out["met_disp"], out["met_gt"], out["met_valid"] = md, mg, mv
for k, v in m.items():
    out["met_" + k.replace(" ", "_").replace(".", "p")] = np.asarray(v)

np.savez_compressed(os.path.join(HERE, "frontend_cases.npz"), **out)
print("wrote", len(out), "arrays")
