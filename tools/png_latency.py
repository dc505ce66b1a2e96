"""Device PNG decode latency: a batch of KITTI-sized 16-bit disparity PNGs (PIL-encoded), one workgroup per file."""
import io, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from vppstereo_amd.engine import Engine
eng = Engine()
rng = np.random.default_rng(0)
files = []
for i in range(32):
    d = (rng.uniform(1, 200, (375, 1242)) * 256).astype(np.uint16)
    d[rng.random(d.shape) > 0.2] = 0
    b = io.BytesIO(); Image.fromarray(d).save(b, format="PNG"); files.append(b.getvalue())
for n in (1, 8, 32):
    eng.png_decode(files[:n], 375, 1242); torch.cuda.synchronize()
    t = time.perf_counter(); eng.png_decode(files[:n], 375, 1242); torch.cuda.synchronize(); dt = time.perf_counter() - t
    t = time.perf_counter()
    for f in files[:n]: np.array(Image.open(io.BytesIO(f))) / 256.0
    dh = time.perf_counter() - t
    print(f"n={n}: device {dt*1e3:.1f} ms ({dt*1e3/n:.2f} ms/file, {sum(map(len, files[:n]))/1e6:.2f} MB compressed); host PIL {dh*1e3:.1f} ms")
