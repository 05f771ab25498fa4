"""ctypes front-end of the CPU oracle (``surge_fold_oracle.c``).

TEST INFRASTRUCTURE ONLY — importable from ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg, never from ``surge_amd/``.  See the C file's header for
the parity status (pinned on the reference specs' explicit values; MurmurHash3 and play-json
text are "parity unpinned").
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

import numpy as np

from surge_amd.schema import (
    CSchema,
    DEFAULT_ALGEBRA,
    EVENT_DTYPE,
    STATE_DTYPE,
    EventAlgebra,
)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib: Optional[ctypes.CDLL] = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "surge_fold_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "surge_replay.h")
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
        L.oracle_fold_csr.argtypes = [ctypes.POINTER(CSchema), vp, i64, vp, vp, vp]
        L.oracle_fold_csr.restype = i32
        L.oracle_fold_csr_v2.argtypes = [vp, vp, i64, vp, vp, vp]
        L.oracle_fold_csr_v2.restype = i32
        L.oracle_fold_csr_mt.argtypes = [ctypes.POINTER(CSchema), vp, i64, vp, vp, vp, i32]
        L.oracle_fold_csr_mt.restype = i32
        L.oracle_fold_csr_mt_reps.argtypes = [ctypes.POINTER(CSchema), vp, i64, vp, vp, vp, i32, i32]
        L.oracle_fold_csr_mt_reps.restype = i32
        L.oracle_handle_event.argtypes = [ctypes.POINTER(CSchema), vp, vp, vp]
        L.oracle_handle_event.restype = i32
        L.oracle_murmur3_string_hash.argtypes = [vp, i64]
        L.oracle_murmur3_string_hash.restype = i32
        L.oracle_partition_for_key.argtypes = [vp, i64, i32]
        L.oracle_partition_for_key.restype = i32
        L.oracle_partition_by_up_to_colon.argtypes = [vp, i64]
        L.oracle_partition_by_up_to_colon.restype = i64
        L.oracle_partition_hash_batch.argtypes = [vp, vp, i64, i32, i32, vp]
        L.oracle_partition_hash_batch.restype = i32
        L.oracle_murmur3_x86_32.argtypes = [ctypes.c_char_p, i64, ctypes.c_uint32]
        L.oracle_murmur3_x86_32.restype = ctypes.c_uint32
        L.oracle_counter_state_json.argtypes = [ctypes.c_char_p, i32, i32, ctypes.c_char_p, i64]
        L.oracle_counter_state_json.restype = i64
        _lib = L
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def fold_csr(
    seg_off: np.ndarray,
    events: np.ndarray,
    init_state: Optional[np.ndarray] = None,
    algebra: EventAlgebra = DEFAULT_ALGEBRA,
    threads: int = 1,
) -> np.ndarray:
    """``events[seg_off[a]:seg_off[a+1]].foldLeft(init[a])(handleEvent)`` for every aggregate."""
    seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
    events = np.ascontiguousarray(events, dtype=EVENT_DTYPE)
    n_agg = seg_off.shape[0] - 1
    assert n_agg >= 0 and seg_off[n_agg] <= events.shape[0]
    if init_state is not None:
        init_state = np.ascontiguousarray(init_state, dtype=STATE_DTYPE)
        assert init_state.shape[0] == n_agg
    out = np.zeros(n_agg, dtype=STATE_DTYPE)
    sc = algebra.to_c()
    if threads <= 1:
        rc = lib().oracle_fold_csr(ctypes.byref(sc), _ptr(seg_off), n_agg, _ptr(events), _ptr(init_state), _ptr(out))
    else:
        rc = lib().oracle_fold_csr_mt(
            ctypes.byref(sc), _ptr(seg_off), n_agg, _ptr(events), _ptr(init_state), _ptr(out), threads
        )
    if rc != 0:
        raise RuntimeError(f"oracle_fold_csr failed: {rc}")
    return out


def fold_csr_v2(seg_off, events, algebra, init_state=None) -> np.ndarray:
    """The sequential fold under an ABI v2 ``SlotAlgebra``; returns ``n_agg x 64`` raw bytes viewed with the
    algebra's ``state_dtype()``."""
    seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
    events = np.ascontiguousarray(events, dtype=EVENT_DTYPE)
    n_agg = seg_off.shape[0] - 1
    out = np.zeros(n_agg, dtype=algebra.state_dtype())
    if init_state is not None:
        init_state = np.ascontiguousarray(init_state)
        assert init_state.nbytes == n_agg * 64
    sc = algebra.to_c()
    rc = lib().oracle_fold_csr_v2(ctypes.byref(sc), _ptr(seg_off), n_agg, _ptr(events), _ptr(init_state), _ptr(out))
    if rc != 0:
        raise RuntimeError(f"oracle_fold_csr_v2 failed: {rc}")
    return out


def fold_csr_repeated(seg_off, events, threads: int, reps: int, algebra: EventAlgebra = DEFAULT_ALGEBRA) -> np.ndarray:
    """``reps`` folds per host thread over its share of the aggregates (timing helper: no thread start per pass)."""
    seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
    events = np.ascontiguousarray(events, dtype=EVENT_DTYPE)
    n_agg = seg_off.shape[0] - 1
    out = np.zeros(n_agg, dtype=STATE_DTYPE)
    sc = algebra.to_c()
    rc = lib().oracle_fold_csr_mt_reps(ctypes.byref(sc), _ptr(seg_off), n_agg, _ptr(events), None, _ptr(out), threads, reps)
    if rc != 0:
        raise RuntimeError(f"oracle_fold_csr_mt_reps failed: {rc}")
    return out


def handle_event(state: np.ndarray, event: np.ndarray, algebra: EventAlgebra = DEFAULT_ALGEBRA) -> np.ndarray:
    """One ``handleEvent`` step on a 1-element state array and a 1-element event array."""
    st = np.ascontiguousarray(state, dtype=STATE_DTYPE).reshape(1)
    ev = np.ascontiguousarray(event, dtype=EVENT_DTYPE).reshape(1)
    out = np.zeros(1, dtype=STATE_DTYPE)
    sc = algebra.to_c()
    lib().oracle_handle_event(ctypes.byref(sc), _ptr(st), _ptr(ev), _ptr(out))
    return out


def _utf16(s: str) -> np.ndarray:
    return np.frombuffer(s.encode("utf-16-le"), dtype=np.uint16)


def murmur3_string_hash(s: str) -> int:
    u = np.ascontiguousarray(_utf16(s))
    return int(lib().oracle_murmur3_string_hash(_ptr(u) if u.size else None, u.size))


def murmur3_x86_32(data: bytes, seed: int = 0) -> int:
    """Appleby's MurmurHash3_x86_32 over bytes, from the same primitives as ``murmur3_string_hash``."""
    return int(lib().oracle_murmur3_x86_32(data, len(data), seed & 0xFFFFFFFF))


def partition_for_key(key: str, n_partitions: int) -> int:
    """``KafkaPartitionProvider.partitionForKey``: the WHOLE string (KafkaPartitioner.scala:8)."""
    u = np.ascontiguousarray(_utf16(key))
    return int(lib().oracle_partition_for_key(_ptr(u) if u.size else None, u.size, n_partitions))


def partition_by_up_to_colon(key: str) -> str:
    """``PartitionStringUpToColon.partitionBy`` (KafkaPartitioner.scala:38-42)."""
    u = np.ascontiguousarray(_utf16(key))
    n = int(lib().oracle_partition_by_up_to_colon(_ptr(u) if u.size else None, u.size))
    return u[:n].tobytes().decode("utf-16-le")


def partition_hash_batch(utf16: np.ndarray, str_off: np.ndarray, n_partitions: int, up_to_colon: bool = False) -> np.ndarray:
    utf16 = np.ascontiguousarray(utf16, dtype=np.uint16)
    str_off = np.ascontiguousarray(str_off, dtype=np.int64)
    n = str_off.shape[0] - 1
    out = np.zeros(n, dtype=np.int32)
    rc = lib().oracle_partition_hash_batch(_ptr(utf16), _ptr(str_off), n, n_partitions, 1 if up_to_colon else 0, _ptr(out))
    if rc != 0:
        raise RuntimeError("oracle_partition_hash_batch failed")
    return out


def counter_state_json(aggregate_id: str, count: int, version: int) -> bytes:
    cap = 6 * len(aggregate_id.encode("utf-8")) + 96
    buf = ctypes.create_string_buffer(cap)
    n = lib().oracle_counter_state_json(aggregate_id.encode("utf-8"), count, version, buf, cap)
    if n < 0:
        raise RuntimeError("oracle_counter_state_json: buffer too small")
    return buf.raw[:n]


def play_json_double_text(x: float) -> str:
    """The JSON text play-json 2.9.2 writes for a Scala ``Double`` — oracle-side restatement, independent of the
    product's Ryu implementation (``surge_amd/csrc/f64_text.h`` states the rule and its sources):

    * digits = ``java.lang.Double.toString``'s: the shortest decimal that rounds back to ``x`` (Python's ``repr`` is David
      Gay's shortest round-trip conversion — the third-party pin for the digits), except that a ONE-digit shortest
      decimal is replaced by the closest TWO-digit decimal that rounds back (exact rational arithmetic below);
    * ``JsNumber(BigDecimal(...))`` -> ``stripTrailingZeros`` -> ``toPlainString`` inside (1E-10, 1E20), else
      ``toString`` -> re-read -> ``BigDecimal.toString``.
    ``""`` for NaN / infinities (``new BigDecimal("NaN")`` throws: the reference's ``writeState`` fails)."""
    from fractions import Fraction

    if x != x or x in (float("inf"), float("-inf")):
        return ""
    if x == 0:
        return "0"
    mant, _, ex = repr(abs(x)).partition("e")
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0")
    e_last = (int(ex) if ex else 0) - len(fp)  # exponent of the last digit
    stripped = digits.rstrip("0")
    e_last += len(digits) - len(stripped)
    digits = stripped
    if len(digits) == 1:
        exact = Fraction(abs(x))
        k = e_last - 1  # two-digit decimals around x are multiples of 10^k with x / 10^k in [10, 100)
        while exact / Fraction(10) ** k >= 100:
            k += 1
        while exact / Fraction(10) ** k < 10:
            k -= 1
        unit = Fraction(10) ** k
        q = exact / unit
        lo = q.numerator // q.denominator
        cands = sorted((abs(exact - c * unit), c % 2, c) for c in (lo, lo + 1) if float(c * unit) == abs(x))
        c = cands[0][2]
        digits, e_last = str(c), k
        stripped = digits.rstrip("0")
        e_last += len(digits) - len(stripped)
        digits = stripped
    n = len(digits)
    a = e_last + n - 1  # adjusted exponent
    if a >= 20 or a < -6:
        s = digits[0] + ("." + digits[1:] if n > 1 else "") + "E" + ("-" if a < 0 else "+") + str(abs(a))
    elif a >= 0:
        s = digits[: a + 1].ljust(a + 1, "0") + ("." + digits[a + 1:] if n > a + 1 else "")
    else:
        s = "0." + "0" * (-a - 1) + digits
    return ("-" if x < 0 else "") + s
