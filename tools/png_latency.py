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
# what the filters cost: the same image with one filter type forced on every row (hand-rolled writer of the tests)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_png as T
d = (np.linspace(256, 40000, 375 * 1242).reshape(375, 1242) + rng.integers(0, 300, (375, 1242))).astype(np.uint16)
for ft in (0, 1, 2, 3, 4):
    b = T._raw_png(d, types=(ft,), extra=False)
    eng.png_decode([b], 375, 1242); torch.cuda.synchronize()
    t = time.perf_counter(); eng.png_decode([b], 375, 1242); torch.cuda.synchronize()
    print(f"filter {ft}: {(time.perf_counter() - t) * 1e3:.1f} ms for one file of {len(b) / 1e6:.2f} MB")
# a lone 64-thread wave does not make the GPU leave its idle clock: the same decode while another stream keeps the chip busy
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device=eng.device, dtype=torch.bfloat16)
for n in (1, 32):
    with torch.cuda.stream(side):
        for _ in range(60):
            a @ a
    time.sleep(0.05)
    t = time.perf_counter(); eng.png_decode(files[:n], 375, 1242); torch.cuda.current_stream().synchronize(); dt = time.perf_counter() - t
    torch.cuda.synchronize()
    print(f"n={n} next to a busy stream: device {dt*1e3:.1f} ms ({dt*1e3/n:.2f} ms/file)")
