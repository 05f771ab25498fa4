"""surge_amd — MI355X-native aggregate-replay engine for UltimateSoftware/surge's state-reconstruction path.

One hot path only: ``events.foldLeft(state)(handleEvent)`` per aggregate, re-expressed
as a segmented fold over a CSR-packed event log and run by hand-written HIP kernels behind the C ABI in
``include/surge_replay.h``.  This package is the host-side mirror of the reference's plugin surface.
"""
from .schema import (  # noqa: F401
    ALGO_AUTO,
    ALGO_FIXED,
    ALGO_FLAT,
    ALGO_ROWS,
    DEFAULT_ALGEBRA,
    EVENT_DTYPE,
    STATE_DTYPE,
    EventAlgebra,
)

__all__ = [
    "ALGO_AUTO",
    "ALGO_FIXED",
    "ALGO_FLAT",
    "ALGO_ROWS",
    "DEFAULT_ALGEBRA",
    "EVENT_DTYPE",
    "STATE_DTYPE",
    "EventAlgebra",
]
