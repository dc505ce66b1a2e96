"""tools/fuzz_long.py [first_seed [n]]: the seeded differential tests of tests/test_gpu_fuzz.py and tests/test_gpu_fused_fuzz.py over
seeds the test suite does not run (a one-off soak after kernel changes; every failure prints its seed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
import tests.test_gpu_fuzz as fz
import tests.test_gpu_fused_fuzz as ff
fails = []
t0 = time.time()
for s in range(first, first + n):
    for name, fn in (("vpp", fz.test_vpp_random_parameters), ("rsgm", fz.test_compute_rsgm_random_parameters)):
        try:
            fn(s)
        except Exception as e:  # noqa
            fails.append((name, s, str(e)[:200]))
_fx = getattr(ff.engines, "__wrapped__", None) or getattr(getattr(ff.engines, "__pytest_wrapped__", None), "obj", None)
engines = _fx() if _fx else None
if engines is None:
    print("could not unwrap the engines fixture: fused fuzz skipped")
if engines is not None:
    for s in range(first, first + n):
        try:
            ff.test_fused_layout_equals_eight_path_layout(engines, s)
        except Exception as e:  # noqa
            fails.append(("fused", s, str(e)[:200]))
        try:
            ff.test_batched_vpp_with_a_mask_random_parameters(engines, s)
        except Exception as e:  # noqa
            fails.append(("batched_vpp", s, str(e)[:200]))
import numpy as np
import tests.test_gpu_post as tp
from vppstereo_amd.engine import Engine
_eng = Engine()
for s in range(first, first + n):
    rng = np.random.default_rng(s)
    h, w = int(rng.integers(5, 160)), int(rng.integers(5, 330))
    lo = int(rng.integers(1, 12))
    maps = [tp._blocks(h, w, rng, lo, lo + int(rng.integers(2, 30))) for _ in range(int(rng.integers(1, 4)))]
    try:
        tp._run(_eng, maps, h, w, rng, subpixel=bool(rng.integers(2)), flip=float(rng.choice([0.0, 0.05, 0.3])))
    except Exception as e:  # noqa
        fails.append(("post", s, str(e)[:200]))
print("seeds %d..%d: %d failures in %.0f s" % (first, first + n - 1, len(fails), time.time() - t0), flush=True)
for f in fails[:20]:
    print(f)
