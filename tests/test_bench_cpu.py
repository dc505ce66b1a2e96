"""CPU checks of bench.py's N-rank launcher and frame plan (no GPU): `--gpus 2` without a launcher must start two
rank processes itself, the ranks must see each other (gloo) and own disjoint, seed-consistent frame blocks."""
import json
import os
import subprocess
import sys

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_plan_is_a_partition_with_sharding_independent_seeds():
    for world in (1, 2, 3, 8):
        plans = [bench.frame_plan(32, r, world) for r in range(world)]
        assert all(p["n_total"] == 32 * world for p in plans)
        assert plans[0]["lo"] == 0 and plans[-1]["hi"] == 32 * world
        for a, b in zip(plans, plans[1:]):
            assert a["hi"] == b["lo"]
        # frame f draws from srand(1 + f) whatever the split
        assert all(p["seed0"] == 1 + p["lo"] for p in plans)


def test_gpus_2_self_spawns_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "5", "--spawn-check"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["frames_per_step"] == 10
    assert line["plan"] == [[0, 5, 1], [5, 10, 6]]


def test_failed_rank_gives_nonzero_exit():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    # without a GPU every rank of a real run exits non-zero ("needs an MI355X"); the parent must report that
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--cpu-frames", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
