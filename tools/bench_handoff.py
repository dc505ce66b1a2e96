"""Roofline numbers of the hand-off kernels at the sizes the reference runs them:
PSMNet volume for a 544x960 (padded 540x960) pair, maxdisp 192, 32 feature channels
(psmnet.py:157-197: [B,64,48,136,240] float32 = 401 MB per frame, write-bound), RAFT-Stereo
correlation modulation ([B,136,240,1,240]), KITTI payload decode.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vppstereo_amd.engine import Engine  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    eng = Engine()
    dev = eng.device
    B, C, H4, W4, H, W, maxdisp = 4, 32, 136, 240, 544, 960, 192
    g = torch.Generator(device="cpu").manual_seed(1)
    fl = torch.randn((B, C, H4, W4), generator=g).to(dev)
    fr = torch.randn((B, C, H4, W4), generator=g).to(dev)
    valid = (torch.rand((B, 1, H, W), generator=g) < 0.03).float().to(dev)
    hints = (torch.rand((B, 1, H, W), generator=g) * 190 + 1).to(dev) * valid
    out = {}
    vol_bytes = B * 2 * C * (maxdisp // 4) * H4 * W4 * 4
    ms = timeit(lambda: eng.psmnet_cost_volume(fl, fr, maxdisp, hints, valid))
    out["psmnet_cost_volume"] = {"ms": round(ms, 4), "GB_written": round(vol_bytes / 1e9, 3),
                                 "write_GBps": round(vol_bytes / ms / 1e6, 1), "frames": B}

    def torch_ref():  # what the reference executes (psmnet.py:157-197), same device
        cost = torch.zeros((B, 2 * C, maxdisp // 4, H4, W4), device=dev)
        for i in range(maxdisp // 4):
            if i > 0:
                cost[:, :C, i, :, i:] = fl[:, :, :, i:]
                cost[:, C:, i, :, i:] = fr[:, :, :, :-i]
            else:
                cost[:, :C, i] = fl
                cost[:, C:, i] = fr
        h = torch.nn.functional.interpolate(hints, size=[H // 4, W // 4], mode="nearest").squeeze(1)
        v = torch.nn.functional.interpolate(valid, size=[H // 4, W // 4], mode="nearest").squeeze(1)
        h = h * v / 4.0
        h = h.unsqueeze(1).unsqueeze(2).expand(-1, 2 * C, maxdisp // 4, -1, -1)
        v = v.unsqueeze(1).unsqueeze(2).expand(-1, 2 * C, maxdisp // 4, -1, -1)
        d = torch.linspace(0, maxdisp // 4 - 1, maxdisp // 4, device=dev).view(1, 1, -1, 1, 1).expand(B, 2 * C, -1, H4, W4)
        return cost * ((1 - v) + v * 10.0 * torch.exp(-(d - h) ** 2 / (2 * 0.25 ** 2)))
    ms_t = timeit(torch_ref, iters=3, warm=1)
    out["psmnet_cost_volume"]["torch_eager_ms"] = round(ms_t, 3)
    err = (eng.psmnet_cost_volume(fl, fr, maxdisp, hints, valid) - torch_ref()).abs().max().item()
    out["psmnet_cost_volume"]["max_abs_diff_vs_torch"] = err

    corr = torch.randn((B, H4, W4, 1, W4), generator=g).to(dev)
    ms = timeit(lambda: eng.raft_corr_modulate_(corr, hints, valid))
    out["raft_corr_modulate"] = {"ms": round(ms, 4), "volume_GB": round(corr.numel() * 4 / 1e9, 3), "frames": B,
                                 "note": "in place, only rows with a hint are touched"}
    png = torch.randint(0, 65536, (64, 375, 1242), generator=g).to(torch.int16).to(dev)
    ms = timeit(lambda: eng.kitti_disp_decode(png))
    nb = png.numel() * (2 + 4 + 1)
    out["kitti_disp_decode"] = {"ms": round(ms, 4), "GBps": round(nb / ms / 1e6, 1), "frames": 64}
    # cfg 4: VPP only -> network tensors, device resident (17 B/pixel: latency bound)
    import synth
    for Bv in (1, 32):
        b = synth.make_batch(min(Bv, 4), 540, 960, 192, 0.03, seed=1234)
        idx = [i % min(Bv, 4) for i in range(Bv)]
        lt = torch.from_numpy(np.ascontiguousarray(b["left"][idx])).to(dev)
        rt = torch.from_numpy(np.ascontiguousarray(b["right"][idx])).to(dev)
        ht = torch.from_numpy(np.ascontiguousarray(b["hints"][idx])).to(dev)

        def vpp_to_net():
            lv, rv = eng.vpp(lt, rt, ht, seed=1)
            return eng.to_network_input(lv, torch.bfloat16), eng.to_network_input(rv, torch.bfloat16)
        ms = timeit(vpp_to_net)
        out[f"vpp_to_bf16_nchw_B{Bv}"] = {"ms": round(ms, 4), "us_per_frame": round(ms * 1e3 / Bv, 1)}
    out["device"] = eng.ctx.device_name
    print(json.dumps(out))


if __name__ == "__main__":
    main()
