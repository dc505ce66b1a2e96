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


def test_gpus_8_self_spawns_eight_ranks_with_the_numa_path_and_the_prediction():
    """`--gpus 8 --spawn-check` on a box without GPUs: eight gloo ranks, a partition of 8 x 4 frames with sharding-independent
    seeds, the NUMA pinning path taken (and harmless) on every rank, and the written-down prediction in the line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--batch", "4", "--spawn-check"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["frames_per_step"] == 32
    assert line["plan"] == [[4 * i, 4 * i + 4, 1 + 4 * i] for i in range(8)]
    assert len(line["pinned"]) == 8 and "pinned" in line["pin_rank0"]
    pw = {p["n_gpus"]: p for p in line["predicted"]["per_world"]}
    assert set(pw) == {1, 2, 4, 8}


def test_uneven_shards_30_frames_over_8_ranks():
    """A total that the ranks do not divide: contiguous blocks whose sizes differ by at most one, seeds by global frame index,
    and the gather's padding arithmetic (every rank sends ceil(30 / 8) frames, the root cuts the padding off)."""
    from vppstereo_amd import dist as vdist
    blocks = [vdist.shard_range(30, r, 8) for r in range(8)]
    assert blocks[0][0] == 0 and blocks[-1][1] == 30
    assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
    sizes = [hi - lo for lo, hi in blocks]
    assert sizes == [4, 4, 4, 4, 4, 4, 3, 3] and max(sizes) == (30 + 7) // 8
    assert [vdist.frame_seed(7, lo) for lo, _ in blocks] == [7 + lo for lo, _ in blocks]
    assert vdist.frame_seed(0xFFFFFFFF, 3) == 2          # srand() takes 32 bits


def test_scaling_prediction_is_what_design_section_10_states():
    p = bench.predict_scaling(8.25, 32, 540, 960, 192)
    rows = {r["n_gpus"]: r for r in p["per_world"]}
    assert rows[1]["efficiency"] == 1.0 and rows[1]["gather_bytes_into_rank0"] == 0
    assert rows[8]["gather_bytes_into_rank0"] == 7 * 32 * 540 * 960 * 4          # 7 x 66 MB
    assert 0.42 < rows[8]["gather_ms_per_link"] < 0.45 and rows[8]["gather_exposed_ms"] == 0.0
    assert abs(rows[8]["ms_per_step"] - 8.25 * 1.04) < 0.01 and abs(rows[2]["ms_per_step"] - 8.25 * (1 + 0.04 / 7)) < 0.01
    assert rows[8]["efficiency"] >= p["near_linear_means_efficiency_at_8_of_at_least"]
    assert rows[2]["efficiency"] > rows[4]["efficiency"] > rows[8]["efficiency"]


def test_all_cores_simd_cpu_leg_counts_its_frames_and_times_only_the_compute():
    """`cpu_baseline_simd_all_cores`: one process per core, released together after their frames exist; the wall clock runs from the
    first start to the last end and the figure is frames x cells over it."""
    r = bench.cpu_baseline_simd_parallel(frames_per_process=1)
    n = r["cores"]
    assert n >= 1 and r["unit"] == "Mdisparities/s" and r["kind"] == "port"
    assert f"{n} frames: 1 on each of {n} processes" in r["sample"]
    per_core = 540 * 960 * 192 / 1e6 / r["s_per_frame_per_core"]
    assert 0.3 * n * per_core < r["value"] <= 1.05 * n * per_core     # between badly skewed and perfectly parallel
    import oracle
    assert not oracle.get_simd()                                         # the parent's checker stays scalar
