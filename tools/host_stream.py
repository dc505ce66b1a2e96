#!/usr/bin/env python3
"""tools/host_stream.py [out.json] [--frames N]: ms per frame of vppstereo_amd.pipeline.FrameStream -- frames start in PAGEABLE numpy
arrays (a pool of distinct frames cycled through, as a DataLoader would hand them over), go through push / pop one at a time,
results land in fresh numpy arrays -- for 540x960x192 (3 % hints) and 375x1242x192 (5 %), occlusion mask on the way, next to the
same frames through `run_frame` (one synchronous call per frame) and the PCIe floor of the bytes that cross (both directions
overlap: the larger one counts).  bench.py reports the same figures as `host_stream` (never as `value`)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth


def measure(H, W, D, p, n_frames, batch=None, depth=3, copy_threads=-1, pool=8, with_run_frame=True):
    from vppstereo_amd import pipeline
    frames = []
    for f in range(pool):
        fr = synth.make_frame(H, W, D, p, seed=900 + f)
        frames.append((fr["left"], fr["right"], fr["hints"]))
    kw = dict(maskocc=True, rsgm_kw=dict(dmax=D))
    res = {"shape": [H, W, D], "hint_density": p, "frames": n_frames}
    with pipeline.FrameStream(H, W, 3, batch=batch, depth=depth, seed=1, copy_threads=copy_threads, **kw) as fs:
        res["batch"], res["depth"] = fs.batch, fs.depth
        gen = ((frames[i % pool]) for i in range(fs.batch * 3))
        for _ in fs.run(gen):          # warm-up: workspace, ring, first-touch of everything
            pass
    with pipeline.FrameStream(H, W, 3, batch=batch, depth=depth, seed=1, copy_threads=copy_threads, **kw) as fs:
        for _ in fs.run((frames[i % pool]) for i in range(fs.batch * 2)):
            pass
        # where the host's time goes: seconds inside push (the copy into the ring + a batch's submission) and inside the native pop
        # (waiting for the batch + the copy out), the rest is Python
        tp = [0.0, 0.0]
        push0, pop0 = fs.push, fs._pop_native

        def push1(*a):
            t = time.perf_counter(); push0(*a); tp[0] += time.perf_counter() - t

        def pop1():
            t = time.perf_counter(); r = pop0(); tp[1] += time.perf_counter() - t
            return r
        fs.push, fs._pop_native = push1, pop1
        t0 = time.perf_counter()
        n = 0
        acc = 0.0
        for d in fs.run((frames[i % pool]) for i in range(n_frames)):
            n += 1
            acc += float(d[0, 0])      # the caller touches the result
        dt = time.perf_counter() - t0
        assert n == n_frames
        res["reruns"] = fs.counts()[3]
        res["host_ms_per_frame"] = {"push": round(tp[0] / n_frames * 1e3, 4), "pop": round(tp[1] / n_frames * 1e3, 4),
                                    "python_rest": round((dt - tp[0] - tp[1]) / n_frames * 1e3, 4)}
    res["stream_ms_per_frame"] = round(dt / n_frames * 1e3, 4)
    res["stream_Mdisp_per_s"] = round(H * W * D / (dt / n_frames) / 1e6, 1)
    up, down = H * W * (3 + 3 + 4), H * W * 4
    res["pcie_bytes_per_frame"] = {"up": up, "down": down}
    res["pcie_floor_ms_per_frame"] = round(max(up, down) / 55e9 * 1e3, 4)     # ~55 GB/s per direction, PCIe 5 x16 in practice
    if with_run_frame:
        m = min(n_frames, 24)
        for i in range(3):
            pipeline.run_frame(*frames[i % pool], **kw)
        t0 = time.perf_counter()
        for i in range(m):
            pipeline.run_frame(*frames[i % pool], **kw)
        res["run_frame_ms_per_frame"] = round((time.perf_counter() - t0) / m * 1e3, 4)
    return res


if __name__ == "__main__":
    n = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 256
    out = {"what": "FrameStream: pageable numpy in, numpy out, one frame at a time (ms per frame, PCIe inclusive)", "runs": []}
    for (H, W, D, p) in ((540, 960, 192, 0.03), (375, 1242, 192, 0.05)):
        for thr in (-1, 1, 2, 8):
            r = measure(H, W, D, p, n, copy_threads=thr, with_run_frame=(thr == -1))
            r["copy_threads"] = thr
            out["runs"].append(r)
            print(json.dumps(r), flush=True)
    path = [a for a in sys.argv[1:] if a.endswith(".json")]
    if path:
        json.dump(out, open(path[0], "w"), indent=1)
