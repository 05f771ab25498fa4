"""Fixed-width layouts and the event algebra (mirror of ``include/surge_replay.h``).

The reference has no fixed-width format: aggregates are JSON and ``handleEvent`` is
arbitrary JVM code (``modules/command-engine/scaladsl/src/main/scala/surge/scaladsl/command/
CommandModels.scala:14``).  A plugin that wants GPU replay declares, beside its
``handleEvent``, one 32-bit descriptor per event type; this module holds the constants,
the ctypes/numpy views of the 16-byte event and 64-byte state, and the built-in schema
that restates the reference's own fixtures (Counter:
``.../scaladsl/src/test/scala/surge/scaladsl/TestBoundedContext.scala:77-89``; BankAccount:
``modules/surge-docs/src/test/scala/docs/command/BankAccountCommandModel.scala:81-86``).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Dict, Tuple, Union, Sequence

import numpy as np

ABI_VERSION = 1
STATE_SIZE = 64
EVENT_SIZE = 16
MAX_EVENT_TYPES = 16

STATE_PRESENT = 1
STATE_POISONED = 2

CLS_MATERIALIZE = 0
CLS_REQUIRE = 1
CLS_CREATE = 2
CLS_DELETE = 3
CLS_MASK = 3
D_POISON = 1 << 2
D_COUNT_ADD = 1 << 4
D_COUNT_SUB = 2 << 4
D_COUNT_SET = 3 << 4
D_COUNT_MASK = 3 << 4
D_VERSION_SET = 1 << 6
D_SUM_ADD = 1 << 8
D_SUM_SUB = 2 << 8
D_SUM_MASK = 3 << 8
D_BALANCE_SET = 1 << 10
D_MIN_ARG = 1 << 11
D_MAX_ARG = 1 << 12
D_EVCOUNT_INC = 1 << 13

EVT_NOOP = 0
EVT_INC = 1
EVT_DEC = 2
EVT_CREATE = 3
EVT_SET_BALANCE = 4
EVT_DELETE = 5
EVT_THROW = 6

ALGO_AUTO = 0
ALGO_FIXED = 1
ALGO_FLAT = 2
ALGO_ROWS = 3
ALGO_SORTED = 4
ALGO_CHUNKED = 5
ALGO_TILED = 7
ALGO_SHORT = 8

INT32_MAX = 2**31 - 1
INT32_MIN = -(2**31)

#: numpy view of ``surge_event16`` (payload as raw 64 bits; ``arg`` is its low word).
EVENT_DTYPE = np.dtype([("type", "<i4"), ("seq", "<i4"), ("raw", "<u8")])
#: numpy view of ``surge_state64``.
STATE_DTYPE = np.dtype(
    [
        ("count", "<i4"),
        ("version", "<i4"),
        ("sum64", "<i8"),
        ("balance", "<f8"),
        ("min_arg", "<i4"),
        ("max_arg", "<i4"),
        ("event_count", "<u4"),
        ("flags", "<u4"),
        ("reserved", "V24"),
    ]
)
assert EVENT_DTYPE.itemsize == EVENT_SIZE and STATE_DTYPE.itemsize == STATE_SIZE


class CState64(ctypes.Structure):
    _fields_ = [
        ("count", ctypes.c_int32),
        ("version", ctypes.c_int32),
        ("sum64", ctypes.c_int64),
        ("balance", ctypes.c_double),
        ("min_arg", ctypes.c_int32),
        ("max_arg", ctypes.c_int32),
        ("event_count", ctypes.c_uint32),
        ("flags", ctypes.c_uint32),
        ("reserved", ctypes.c_uint8 * 24),
    ]


class CSchema(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_uint32),
        ("state_size", ctypes.c_uint32),
        ("event_size", ctypes.c_uint32),
        ("n_types", ctypes.c_uint32),
        ("desc", ctypes.c_uint32 * MAX_EVENT_TYPES),
        ("default_state", CState64),
    ]


class CStats(ctypes.Structure):
    _fields_ = [
        ("n_aggregates", ctypes.c_int64),
        ("n_events", ctypes.c_int64),
        ("algorithmic_bytes", ctypes.c_int64),
        ("last_fold_kernel_ms", ctypes.c_double),
        ("last_fold_total_ms", ctypes.c_double),
        ("h2d_ms", ctypes.c_double),
        ("last_algo", ctypes.c_int32),
        ("n_tasks", ctypes.c_int32),
        ("n_folds", ctypes.c_int64),
        ("n_poisoned", ctypes.c_int64),
        ("sum_fold_kernel_ms", ctypes.c_double),
        ("timed_folds", ctypes.c_int64),
    ]


class CLayoutInfo(ctypes.Structure):
    """``surge_replay_layout_info_t``: the per-log index of the last SORTED / CHUNKED / TILED fold and its one-off cost."""

    _fields_ = [
        ("algo", ctypes.c_int32),
        ("chunk_events", ctypes.c_int32),
        ("virtual_rows", ctypes.c_int64),
        ("cut_aggregates", ctypes.c_int64),
        ("tiled_bytes", ctypes.c_int64),
        ("padding_events", ctypes.c_int64),
        ("index_build_ms", ctypes.c_double),
        ("relayout_ms", ctypes.c_double),
    ]


class CKernelInfo(ctypes.Structure):
    """``surge_replay_kernel_info_t``: which build of the fold kernels a handle runs."""

    _fields_ = [
        ("specialised", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("compile_ms", ctypes.c_double),
        ("detail", ctypes.c_char * 240),
    ]


assert ctypes.sizeof(CState64) == STATE_SIZE
assert ctypes.sizeof(CSchema) == 16 + 4 * MAX_EVENT_TYPES + STATE_SIZE


@dataclass(frozen=True)
class EventAlgebra:
    """Declarative restatement of a model's ``handleEvent`` (one descriptor per event type).

    ``default_*`` are the fields an absent aggregate materialises to — the
    ``State(evt.aggregateId, 0, 0)`` of ``TestBoundedContext.scala:78``.
    """

    desc: Sequence[int]
    default_count: int = 0
    default_version: int = 0
    default_sum64: int = 0
    default_balance: float = 0.0
    default_min_arg: int = INT32_MAX
    default_max_arg: int = INT32_MIN
    default_event_count: int = 0
    names: Sequence[str] = field(default_factory=tuple)

    def __post_init__(self):
        if not 1 <= len(self.desc) <= MAX_EVENT_TYPES:
            raise ValueError(f"an event algebra has 1..{MAX_EVENT_TYPES} event types")

    def to_c(self) -> CSchema:
        s = CSchema()
        s.abi_version = ABI_VERSION
        s.state_size = STATE_SIZE
        s.event_size = EVENT_SIZE
        s.n_types = len(self.desc)
        for i, d in enumerate(self.desc):
            s.desc[i] = int(d) & 0xFFFFFFFF
        d = s.default_state
        d.count = self.default_count
        d.version = self.default_version
        d.sum64 = self.default_sum64
        d.balance = self.default_balance
        d.min_arg = self.default_min_arg
        d.max_arg = self.default_max_arg
        d.event_count = self.default_event_count
        d.flags = STATE_PRESENT
        return s


_COUNTER_EXTRAS = D_MIN_ARG | D_MAX_ARG | D_EVCOUNT_INC

#: The built-in algebra; must equal ``surge_replay_default_schema`` (tested).
DEFAULT_ALGEBRA = EventAlgebra(
    desc=(
        CLS_MATERIALIZE,                                                          # NOOP
        CLS_MATERIALIZE | D_COUNT_ADD | D_VERSION_SET | D_SUM_ADD | _COUNTER_EXTRAS,  # INC
        CLS_MATERIALIZE | D_COUNT_SUB | D_VERSION_SET | D_SUM_SUB | _COUNTER_EXTRAS,  # DEC
        CLS_CREATE | D_BALANCE_SET | D_EVCOUNT_INC,                               # CREATE
        CLS_REQUIRE | D_BALANCE_SET | D_EVCOUNT_INC,                              # SET_BALANCE
        CLS_DELETE,                                                               # DELETE
        D_POISON,                                                                 # THROW
    ),
    names=("NOOP", "INC", "DEC", "CREATE", "SET_BALANCE", "DELETE", "THROW"),
)


# ---- ABI v2: slot schemas (include/surge_replay.h) ---------------------------------------------------------------
SLOT_I32, SLOT_I64, SLOT_F64 = 1, 2, 3
SRC_ARG, SRC_SEQ, SRC_PAYLOAD, SRC_ONE = 0, 1, 2, 3
OP_KEEP, OP_ADD, OP_SUB, OP_SET, OP_MIN, OP_MAX = 0, 1, 2, 3, 4, 5
MAX_SLOTS = 7
ALGO_SLOTS = 6
V2_COUNT_EVENTS = 1


def slot_offset(i: int) -> int:
    """Byte offset of slot ``i`` in the 64-byte state (``SURGE_SLOT_OFFSET``)."""
    return 8 * i if i < 4 else 8 * i + 8


class _CSlotDef(ctypes.Structure):
    _fields_ = [("type", ctypes.c_uint8), ("source", ctypes.c_uint8), ("reserved", ctypes.c_uint8 * 6), ("default_bits", ctypes.c_uint64)]


class CSchemaV2(ctypes.Structure):
    _fields_ = [("abi_version", ctypes.c_uint32), ("state_size", ctypes.c_uint32), ("event_size", ctypes.c_uint32),
                ("n_types", ctypes.c_uint32), ("n_slots", ctypes.c_uint32), ("flags", ctypes.c_uint32),
                ("slot", _CSlotDef * MAX_SLOTS), ("cls", ctypes.c_uint32 * MAX_EVENT_TYPES), ("ops", ctypes.c_uint32 * MAX_EVENT_TYPES)]


@dataclass(frozen=True)
class Slot:
    """One typed 8-byte slot of a v2 state: its value type, where its operand comes from, its default."""

    name: str
    type: int
    source: int
    default: Union[int, float] = 0

    def default_bits(self) -> int:
        if self.type == SLOT_F64:
            return int(np.float64(self.default).view(np.uint64))
        return int(self.default) & (0xFFFFFFFF if self.type == SLOT_I32 else 0xFFFFFFFFFFFFFFFF)


@dataclass(frozen=True)
class SlotAlgebra:
    """ABI v2 declaration of a model's ``handleEvent``: ``slots`` and, per event type, ``(cls, {slot name: op})``."""

    slots: Sequence[Slot]
    types: Sequence[Tuple[int, Dict[str, int]]]
    count_events: bool = False
    names: Sequence[str] = field(default_factory=tuple)

    def __post_init__(self):
        if not 1 <= len(self.slots) <= MAX_SLOTS or not 1 <= len(self.types) <= MAX_EVENT_TYPES:
            raise ValueError(f"a slot algebra has 1..{MAX_SLOTS} slots and 1..{MAX_EVENT_TYPES} event types")

    def to_c(self) -> CSchemaV2:
        s = CSchemaV2()
        s.abi_version, s.state_size, s.event_size = 2, STATE_SIZE, EVENT_SIZE
        s.n_types, s.n_slots, s.flags = len(self.types), len(self.slots), V2_COUNT_EVENTS if self.count_events else 0
        index = {}
        for i, sl in enumerate(self.slots):
            s.slot[i].type, s.slot[i].source, s.slot[i].default_bits = sl.type, sl.source, sl.default_bits()
            index[sl.name] = i
        for t, (cls, ops) in enumerate(self.types):
            s.cls[t] = cls
            word = 0
            for name, op in ops.items():
                word |= (op & 15) << (4 * index[name])
            s.ops[t] = word
        return s

    def state_dtype(self) -> np.dtype:
        """numpy view of a v2 state: one field per slot (at its offset), ``event_count`` and ``flags``."""
        names, formats, offsets = [], [], []
        for i, sl in enumerate(self.slots):
            names.append(sl.name)
            formats.append({SLOT_I32: "<i4", SLOT_I64: "<i8", SLOT_F64: "<f8"}[sl.type])
            offsets.append(slot_offset(i))
        names += ["event_count", "flags"]
        formats += ["<u4", "<u4"]
        offsets += [32, 36]
        return np.dtype({"names": names, "formats": formats, "offsets": offsets, "itemsize": 64})


def empty_states(n: int) -> np.ndarray:
    """``n`` canonical ``None`` states."""
    return np.zeros(n, dtype=STATE_DTYPE)


def make_events(types, seqs, args=None, values=None) -> np.ndarray:
    """Pack parallel arrays into ``surge_event16`` records.

    ``args`` (int32, sign-extended into the low payload word, high word zero) and ``values``
    (float64 bit patterns) are merged: where ``values`` is not NaN-masked out it wins.
    """
    types = np.asarray(types, dtype=np.int32)
    ev = np.zeros(types.shape[0], dtype=EVENT_DTYPE)
    ev["type"] = types
    ev["seq"] = np.asarray(seqs, dtype=np.int32)
    if args is not None:
        ev["raw"] = np.asarray(args, dtype=np.int32).astype(np.uint32).astype(np.uint64)
    if values is not None:
        vals = np.asarray(values, dtype=np.float64)
        use = ~np.isnan(vals) if args is not None else np.ones(vals.shape, dtype=bool)
        ev["raw"][use] = vals.view(np.uint64)[use]
    return ev


def present_mask(states: np.ndarray) -> np.ndarray:
    return (states["flags"] & STATE_PRESENT) != 0


def poisoned_mask(states: np.ndarray) -> np.ndarray:
    return (states["flags"] & STATE_POISONED) != 0
