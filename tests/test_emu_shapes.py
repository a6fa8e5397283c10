"""Seeded random shapes through the whole composite step on the host emulator: batch sizes off the 16-sequence tiles, window
lengths that leave ragged conv tiles and few frames, every head count up to the tile, negatives on and off the 16-wide candidate
tile -- each against the oracle's train step (losses, accuracies, every gradient) and against the stage-wise entry points."""
import random

import pytest

from emu_util import emu
from test_emu_train_step import check_composite_step, frames


def _shapes(n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        B = rng.choice([1, 2, 3, 5])
        L = rng.randint(480, 2300)                             # any window length (--sizeWindow is free in the reference)
        S = frames(L)
        K = rng.randint(1, min(S - 1, 6))
        N = rng.choice([1, 3, 16, 17, 32, 40])
        out.append((B, L, K, N, rng.random() < 0.5))
    return out


# (1, 405, 1, 1): two frames, one window whose only negative is its own positive -- an exactly-zero reference gradient
@pytest.mark.parametrize("B,L,K,N,use_h0", [(1, 405, 1, 1, False)] + _shapes(8, seed=20260928))
def test_composite_step_on_random_shapes_emulated(B, L, K, N, use_h0):
    check_composite_step(emu(), B, L, K, N, use_h0, seed=11)
