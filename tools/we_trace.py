"""tools/we_trace.py [B [H W D]] (experiment build: `tools/build_exp.sh`, VPPX_LIB=tools/bin/libvppx_exp.so): when each wave of a W/E
launch starts and ends and where it runs.  The kernel leaves, per block, the 100 MHz wall clock at its start and end and its HW_ID /
XCC_ID (VPPX_EXP_WE_TRACE=<file>); this script runs one fused call of B frames, re-launches W/E once and prints: blocks per (XCD, CU, SIMD)
-- the placement --, how many blocks start late (after the first block has ended), and the start / duration histogram of whole lines and
of the pieces of cut tail lines.  VPPX_VARIANT=we_whole for the uncut launch."""
import json, os, struct, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
path = os.path.join(tempfile.gettempdir(), "we_trace.bin")
os.environ["VPPX_EXP_WE_TRACE"] = path
import torch
import synth
from vppstereo_amd.engine import Engine

argv = [a for a in sys.argv[1:] if not a.startswith('--')]
B = int(argv[0]) if argv else 16
H, W, D = (int(v) for v in argv[1:4]) if len(argv) > 3 else (540, 960, 192)
eng = Engine()
b = synth.make_batch(min(B, 8), H, W, D, 0.03, seed=1234)
idx = [i % min(B, 8) for i in range(B)]
l, r, h = (torch.from_numpy(np.ascontiguousarray(b[k][idx])).to(eng.device) for k in ("left", "right", "hints"))
for _ in range(2):
    eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", rsgm_kw=dict(dmax=D))
    torch.cuda.synchronize()
if "--instep" in sys.argv:               # the file then holds the W/E launch of the last step's last part, behind its lock-step launch
    eng.set_pipeline(True)
    for _ in range(3):
        eng.vpp_rsgm(l, r, h, g_occ="occlusion_heuristic", rsgm_kw=dict(dmax=D))
    torch.cuda.synchronize()
else:
    eng.time_aggregate_part(1, 3)        # W/E alone, re-launched: the file holds the last launch
raw = np.fromfile(path, dtype=np.uint64)
grid, n_whole, pieces, tail_base, tail_pitch, ntail = (int(v) for v in raw[:6])
t_all = raw[6:].reshape(grid, 3)
live = t_all[:, 1] != 0                      # (blocks that only pad the grid leave zeros)
blk = np.flatnonzero(live)
t = t_all[live]
is_tail = blk >= tail_base if pieces > 1 else np.zeros(blk.size, bool)
piece_of = np.where(is_tail, (blk - tail_base) // max(tail_pitch, 1), -1)
order = np.argsort(is_tail, kind="stable")   # whole lines first, as the rest of the script expects
t, piece_of = t[order], piece_of[order]
n_whole = int((~is_tail).sum())
grid = int(blk.size)
t0 = int(t[:, 0].min())
start = (t[:, 0].astype(np.int64) - t0) / 100.0      # us
end = (t[:, 1].astype(np.int64) - t0) / 100.0
hw = t[:, 2] & np.uint64(0xFFFFFFFF)
xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64)
# HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx950: se_id 3 bits at 13)
simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64)
cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64)
se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
simd_key = key * 4 + simd
per_simd = np.bincount(simd_key)
per_simd = per_simd[per_simd > 0]
res = {"frames": B, "shape": [H, W, D], "grid": grid, "n_whole": n_whole, "pieces": pieces, "launch_us": round(float(end.max()), 1),
       "simds_seen": int(per_simd.size), "cus_seen": int(np.unique(key).size),
       "blocks_per_simd_hist": {int(k): int(v) for k, v in zip(*np.unique(per_simd, return_counts=True))},
       "first_block_ends_us": round(float(end.min()), 1), "blocks_starting_after_that": int((start > end.min()).sum()),
       "start_us_percentiles_whole": [round(float(np.percentile(start[:n_whole], q)), 1) for q in (0, 50, 90, 99, 100)],
       "duration_us_percentiles_whole": [round(float(np.percentile((end - start)[:n_whole], q)), 1) for q in (0, 10, 50, 90, 100)]}
if grid > n_whole:
    res["start_us_percentiles_tail"] = [round(float(np.percentile(start[n_whole:], q)), 1) for q in (0, 50, 90, 100)]
    res["duration_us_percentiles_tail"] = [round(float(np.percentile((end - start)[n_whole:], q)), 1) for q in (0, 10, 50, 90, 100)]
    res["end_us_percentiles_tail"] = [round(float(np.percentile(end[n_whole:], q)), 1) for q in (0, 50, 90, 100)]
    pc = piece_of[n_whole:]
    res["tail_end_us_by_piece"] = [round(float(end[n_whole:][pc == j].mean()), 1) for j in range(max(pieces, 1))]
    res["tail_start_us_by_piece"] = [round(float(start[n_whole:][pc == j].mean()), 1) for j in range(max(pieces, 1))]
# whole lines that share their SIMD with how many others: duration by the SIMD's block count
dur = end - start
cnt_of = np.bincount(simd_key)[simd_key]
res["whole_duration_us_by_blocks_on_its_simd"] = {int(c): round(float(dur[:n_whole][cnt_of[:n_whole] == c].mean()), 1) for c in np.unique(cnt_of[:n_whole])}
per_cu = np.bincount(key)
per_cu = per_cu[per_cu > 0]
res["blocks_per_cu_hist"] = {int(k): int(v) for k, v in zip(*np.unique(per_cu, return_counts=True))}
is_piece = np.arange(grid) >= n_whole
comp = {}
for c in np.unique(cnt_of):
    sel = cnt_of == c
    simds = np.unique(simd_key[sel])
    pieces_on = np.array([int(is_piece[simd_key == sk].sum()) for sk in simds[:400]])
    comp[int(c)] = {int(k): int(v) for k, v in zip(*np.unique(pieces_on, return_counts=True))}
res["pieces_on_a_simd_by_its_block_count"] = comp
# dispatch order against placement: the CU and SIMD of the first 48 blocks of XCD 0 (block ids 0, 8, 16, ...)
if "--order" in sys.argv:
    # (arrays are in block-id order for the whole lines: the first n_whole entries)
    res["first_blocks_xcd0"] = ["%d.%d.%d.%d" % (int(se[i]), int(sh[i]), int(cu[i]), int(simd[i])) for i in range(0, 8 * 160, 8)]
    # per CU of XCD 0: the SIMD sequence of its whole-line blocks in block-id order
    seqs = {}
    for i in range(0, n_whole, 8):
        seqs.setdefault("%d.%d.%d" % (int(se[i]), int(sh[i]), int(cu[i])), []).append(int(simd[i]))
    res["simd_sequence_per_cu_xcd0"] = {k: "".join(map(str, v)) for k, v in list(seqs.items())[:40]}
print(json.dumps(res))
