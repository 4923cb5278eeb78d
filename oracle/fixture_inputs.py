"""Deterministic INPUTS shared by the fixture generator (oracle/make_fixtures.py, build container only) and the
tests that consume its vectors: the synthetic training batch of BASELINE config 3 and the positions of the
gradient samples kept in tests/golden/c3_train_640x360_b2.npz.  Test infrastructure only."""
import numpy as np
import torch

from sfh_amd import synth

GRAD_SAMPLES = 512


def grad_sample_index(numel):
    """positions of a parameter gradient kept in the C3 golden (evenly spread, deterministic)"""
    n = min(numel, GRAD_SAMPLES)
    return np.unique(np.linspace(0, numel - 1, n).astype(np.int64))


def c3_batch(B, H, W, npts, seed=0):
    """ground truth of one synthetic training batch (masks uniform{0..3}, gt POI uniform[0,1])"""
    g = synth._rng(seed, f"trainbatch{H}x{W}")
    nz = (g.uniform(0, 1, (B, npts)) > 0.3).astype(np.float32)
    return {"mask": torch.from_numpy(g.integers(0, 4, (B, H, W)).astype(np.int64)),
            "weight": torch.from_numpy((g.uniform(0, 1, (B,)) + 0.5).astype(np.float32)),
            "poi": torch.from_numpy(g.uniform(0, 1, (B, npts, 2)).astype(np.float32)),
            "nonzeros": torch.from_numpy(nz),
            "num_nonzero": torch.from_numpy(np.maximum(nz.sum(1), 1.0).astype(np.float32))}
