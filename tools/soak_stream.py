#!/usr/bin/env python3
"""tools/soak_stream.py [frames [H W D]]: the frame stream under an erratic caller.  The same sequence of frames goes through
(a) a ring of 2 driven by `run()` (every batch's copy-out right behind its kernels) and (b) rings of 3 and 4 driven by hand: pushes,
flushes at random points (batches of every size), pops of random counts at random times -- so copy-outs are released by the next
submit, by a pop that comes first, by a flush -- and every frame's disparity map must have the same checksum in all of them.
Prints one line per configuration; exits 1 on the first difference (with the frame and the seed)."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    H, W, D = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (540, 960, 192)
    from vppstereo_amd.pipeline import FrameStream
    pool = []
    for f in range(6):
        fr = synth.make_frame(H, W, D, 0.03, seed=4000 + f)
        pool.append((fr["left"], fr["right"], fr["hints"]))
    kw = dict(maskocc=True, rsgm_kw=dict(dmax=D), seed=5)
    with FrameStream(H, W, 3, depth=2, **kw) as fs:
        want = [zlib.crc32(d.tobytes()) for d in fs.run(pool[i % 6] for i in range(n))]
        print(f"reference: ring of 2, run(): {n} frames, batch {fs.batch}, reruns {fs.counts()[3]}", flush=True)
    for depth, seed in ((3, 1), (3, 2), (4, 3)):
        rng = np.random.default_rng(seed)
        got = []
        with FrameStream(H, W, 3, depth=depth, **kw) as fs:
            flushes = pops = 0
            for i in range(n):
                fs.push(*pool[i % 6])
                u = rng.random()
                if u < 0.03:
                    fs.flush()
                    flushes += 1
                if rng.random() < 0.08:
                    for _ in range(int(rng.integers(1, 3 * fs.batch))):
                        r = fs.pop()
                        if r is None:
                            break
                        got.append(zlib.crc32(r.tobytes()))
                        pops += 1
            fs.flush()
            while True:
                r = fs.pop()
                if r is None:
                    break
                got.append(zlib.crc32(r.tobytes()))
            reruns = fs.counts()[3]
        if len(got) != n:
            print(f"ring of {depth}, seed {seed}: {len(got)} results for {n} frames")
            return 1
        bad = [i for i in range(n) if got[i] != want[i]]
        if bad:
            print(f"ring of {depth}, seed {seed}: frame {bad[0]} differs ({len(bad)} in all)")
            return 1
        print(f"ring of {depth}, erratic caller (seed {seed}): {n} frames equal, {flushes} flushes, {pops} early pops, reruns {reruns}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
