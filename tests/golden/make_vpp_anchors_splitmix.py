#!/usr/bin/env python3
"""Full-size VPP anchors whose INPUTS come from the repository's own splitmix64 generator (synth.anchor_inputs_splitmix),
so that the anchor tests can never skip because numpy's Generator stream changed.  Runs only in the build container:
cythonizes /root/reference/vpp_core/vpp_core_opt.pyx in a temporary directory (make_vpp_golden.build_reference), drives
the reference's two scans (vpp_core_opt.pyx:53,133) and stores SHA-256 prefixes of inputs and outputs in
tests/golden/vpp_anchors_splitmix.json.  Numbers only; no reference source."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import synth  # noqa: E402
from make_vpp_golden import build_reference, sha  # noqa: E402


def main():
    v = build_reference()
    H, W, D, p = 540, 960, 192, 0.03
    l, r, g, occ0, occ1 = synth.anchor_inputs_splitmix(H, W, D, p)
    doc = dict(recipe="synth.anchor_inputs_splitmix (SURVEY App. D recipe, splitmix64 inputs)", H=H, W=W, D=D, p=p,
               inputs=dict(l=sha(l), r=sha(r), g=sha(g), occ1=sha(occ1)), cases=[])
    for nm, method, occ in [("rnd_occ0", "rnd", occ0), ("rnd_occ1", "rnd", occ1), ("maxdist_occ1", "maxdist", occ1)]:
        a, b = l.copy(), r.copy()
        v.init_rand(1)
        if method == "rnd":
            n = v.virtual_projection_scan_rnd(a, b, g, W, H, 3, False, 3, 1, 0.4, 0.0, occ, False, True)
        else:
            n = v.virtual_projection_scan_max_dist(a, b, g, W, H, 3, False, 3, 64, 3, 1, 0.4, 0.0, occ, False, True)
        doc["cases"].append(dict(name=nm, n_hints=int(n), l=sha(a), r=sha(b)))
        print(nm, n, sha(a), sha(b))
    a, b = l.copy(), r.copy()
    v.init_rand(3)
    n = v.virtual_projection_scan_rnd(a, b, g, W, H, 3, True, 7, 0, 0.4, 0.25, occ1, False, True)
    doc["cases"].append(dict(name="rnd_w7_uniform_r2l_cocc", n_hints=int(n), l=sha(a), r=sha(b), seed=3))
    with open(os.path.join(HERE, "vpp_anchors_splitmix.json"), "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
