"""Harness counterparts of the reference's ``losses.py`` (rows a20 of SURVEY section 8):
``sample_hints`` (losses.py:5-10) and ``guided_metrics`` (losses.py:13-24).  Plain
torch / numpy like the reference (they are callers of the hot path, not part of it)."""
import numpy as np


def sample_hints(hints, validhints, probability=0.20):
    """Keep each valid hint with the given probability (losses.py:5-10): one torch.rand_like draw per
    element of validhints (so the RNG stream is consumed like the reference does), dropped hints are
    exactly 0 even where the input held inf / NaN.  Works on CPU and device tensors alike."""
    import torch
    keep = torch.rand_like(validhints, dtype=torch.float32) < probability
    sampled_valid = (validhints * keep).float()
    product = hints * sampled_valid
    sampled_hints = torch.where(sampled_valid == 0, torch.zeros_like(product), product)
    return sampled_hints, sampled_valid


def guided_metrics(disp, gt, valid):
    error = np.abs(disp - gt)
    error[valid == 0] = 0
    v = valid > 0
    bad = [(error[v] > t).astype(np.float32).mean() for t in (1., 2., 3., 4.)]
    avgerr = error[v].mean()
    rms = np.sqrt(((disp - gt) ** 2)[v].mean())
    return {'bad 1.0': bad[0], 'bad 2.0': bad[1], 'bad 3.0': bad[2], 'bad 4.0': bad[3], 'avgerr': avgerr, 'rms': rms,
            'errormap': error * v}
