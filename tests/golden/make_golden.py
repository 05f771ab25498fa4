"""Regenerates tests/golden/*.npz.

The reference is JVM-only and cannot be run in this image, and it ships no golden vectors
(SURVEY §4), so these fixtures are produced by the CPU oracle (oracle/surge_fold_oracle.c), which is
itself pinned on the reference specs' explicit values by tests/test_oracle_kat.py.  They freeze
(input log, expected 64-byte states) pairs so that both the oracle and the HIP path are held to the
same bytes from now on.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle  # noqa: E402
from surge_amd import synth  # noqa: E402


def save(name, seg_off, events, init=None, recipe=""):
    """Logs above 150k events store only their sha256 plus the synth recipe that regenerates them
    (the generators are deterministic and tested numpy == torch); smaller ones store the bytes."""
    import hashlib

    expected = oracle.fold_csr(seg_off, events, init)
    big = events.shape[0] > 150_000
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        seg_off=seg_off,
        events=(np.zeros(0, np.uint8) if big else events.view(np.uint8)),
        events_sha256=np.frombuffer(hashlib.sha256(events.tobytes()).digest(), dtype=np.uint8),
        recipe=np.array(recipe),
        init=(init.view(np.uint8) if init is not None else np.zeros(0, np.uint8)),
        expected=expected.view(np.uint8),
    )
    print(name, "aggregates", seg_off.shape[0] - 1, "events", events.shape[0])


def main():
    # C1 of BASELINE.json: 1k aggregates x 100 fixed-width events, Counter types only
    save("c1_counter_1k_x_100", *synth.fixed_log(1000, 100, seed=1, mix=synth.C1_MIX, small_args=True))
    # C2's shape, scaled down: fixed fan-in 256, C2 type mix
    save("c2_shape_512_x_256", *synth.fixed_log(512, 256, seed=2))
    # C3's shape, scaled down: Zipf(1..4096) counts
    save("c3_shape_zipf_1500", *synth.zipf_log(1500, seed=3), recipe="zipf_log(1500, seed=3)")
    # tombstones + throwing events + ragged/empty segments + prior snapshot
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 40, size=4000) * (rng.random(4000) < 0.8)
    so, ev = synth.csr_log(lens, 6, synth.STRESS_MIX)
    prior = oracle.fold_csr(*synth.csr_log(rng.integers(0, 5, size=4000), 7, synth.STRESS_MIX))
    save("stress_ragged_with_snapshot", so, ev, prior)


if __name__ == "__main__":
    main()
