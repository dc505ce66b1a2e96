"""Two processes driving the fused (lock-step) layout on ONE GPU at the same time: the documented unsupported case.
Expected: slower steps, possibly a reported loss of lock step (an exception from the call after the one that lost
it, then the 8-path layout), never a hang and never a wrong result without an error."""
import multiprocessing as mp, sys, time

def worker(rank, q):
    import numpy as np, torch
    sys.path.insert(0, "."); sys.path.insert(0, "tests")
    import synth
    from vppstereo_amd.engine import Engine
    B, H, W, D = 16, 540, 960, 192
    b = synth.make_batch(4, H, W, D, 0.03, seed=1)
    dev = torch.device("cuda:0")
    rep = lambda a: torch.from_numpy(np.concatenate([a] * (B // 4))).to(dev)
    left, right, hints = rep(b["left"]), rep(b["right"]), rep(b["hints"])
    eng = Engine()
    out = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    ref = eng.vpp_rsgm(left, right, hints, seed=1, rsgm_kw=dict(dmax=D)).clone()
    torch.cuda.synchronize()
    q.put(("ready", rank)); time.sleep(1.0)
    errs, bad, t0 = 0, 0, time.perf_counter()
    n = 150
    for i in range(n):
        try:
            eng.vpp_rsgm(left, right, hints, out=out, seed=1, rsgm_kw=dict(dmax=D))
            torch.cuda.synchronize()
            bad += int(not torch.equal(out, ref))
        except Exception as e:  # noqa: BLE001
            errs += 1
            last = str(e)[:80]
    q.put((rank, round((time.perf_counter() - t0) / n * 1e3, 2), "ms/step, errors", errs, "silent mismatches", bad, "layout now", eng.uses_vert()))

if __name__ == "__main__":
    mp.set_start_method("spawn")
    q = mp.Queue()
    ps = [mp.Process(target=worker, args=(r, q)) for r in range(2)]
    for p in ps: p.start()
    for p in ps: p.join(600)
    while not q.empty(): print(q.get())
