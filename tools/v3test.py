import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle, synth
from vppstereo_amd.engine import Engine
eng = Engine()
for (B, H, W, D) in ((4, 50, 150, 192), (4, 100, 400, 192), (8, 33, 200, 192), (4, 20, 24, 192)):
    b = synth.make_batch(B, H, W, D, 0.05, seed=B * H)
    dev = eng.device
    lv = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev); rv = torch.empty_like(lv)
    t0 = time.time()
    out = eng.vpp_rsgm(torch.from_numpy(b["left"]).to(dev), torch.from_numpy(b["right"]).to(dev),
                       torch.from_numpy(b["hints"]).to(dev), l_vpp=lv, r_vpp=rv, seed=11, rsgm_kw=dict(dmax=D))
    torch.cuda.synchronize()
    print("shape", B, H, W, D, "gpu s", round(time.time() - t0, 3), flush=True)
    out, lv, rv = out.cpu().numpy(), lv.cpu().numpy(), rv.cpu().numpy()
    bad = 0
    for f in range(B):
        oracle.init_rand(11 + f)
        lo, ro = oracle.vpp(b["left"][f], b["right"][f], b["hints"][f])
        ref = oracle.compute_rsgm(b["left"][f], lo, ro, dmax=D)
        if not np.array_equal(ref, out[f]):
            bad += 1
            d = np.argwhere(ref != out[f])
            print("  frame", f, "mismatches", len(d), "first", d[:5].tolist(), flush=True)
    print("  ", "OK" if bad == 0 else "BAD", "uses_vert", eng.uses_vert(), flush=True)
