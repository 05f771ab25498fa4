"""GPU-side state encoders: fixed 64-byte states -> the plugin's serialized text, in bulk.

Point reads keep using the plugin's own ``writeState`` on the host (``surge_amd/store.py``); this is for
publishing a whole snapshot (10 M aggregates ≈ 1 GB of JSON) without a per-aggregate host loop.  The
text shape is declared as a template; ``JsonTemplate.counter()`` is the Counter fixture's play-json form
``{"aggregateId":"<id>","count":N,"version":N}``
(``modules/command-engine/scaladsl/src/test/scala/surge/scaladsl/TestBoundedContext.scala:15-16,127-129``).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Sequence, Tuple, Union

import numpy as np

from . import _native
from .replay import ReplayEngine, ReplayError

JP_LITERAL, JP_KEY, JP_I32, JP_U32, JP_I64 = 0, 1, 2, 3, 4
MAX_PARTS = 16


class _CPart(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("field_offset", ctypes.c_uint32), ("lit_off", ctypes.c_uint32),
                ("lit_len", ctypes.c_uint32)]


class CJsonTemplate(ctypes.Structure):
    _fields_ = [("n_parts", ctypes.c_uint32), ("part", _CPart * MAX_PARTS), ("literals", ctypes.c_uint8 * 256)]


@dataclass(frozen=True)
class JsonTemplate:
    """Parts are ``bytes`` literals, the string ``"KEY"``, or ``(kind, state_byte_offset)`` tuples."""

    parts: Sequence[Union[bytes, str, Tuple[int, int]]]

    @staticmethod
    def counter() -> "JsonTemplate":
        return JsonTemplate((b'{"aggregateId":', "KEY", b',"count":', (JP_I32, 0), b',"version":', (JP_I32, 4), b"}"))

    def to_c(self) -> CJsonTemplate:
        t = CJsonTemplate()
        if not 1 <= len(self.parts) <= MAX_PARTS:
            raise ValueError(f"a template has 1..{MAX_PARTS} parts")
        t.n_parts = len(self.parts)
        pool = 0
        for i, p in enumerate(self.parts):
            if isinstance(p, bytes):
                if pool + len(p) > 256:
                    raise ValueError("literal pool exceeds 256 bytes")
                t.part[i].kind, t.part[i].lit_off, t.part[i].lit_len = JP_LITERAL, pool, len(p)
                for b in p:
                    t.literals[pool] = b
                    pool += 1
            elif p == "KEY":
                t.part[i].kind = JP_KEY
            else:
                t.part[i].kind, t.part[i].field_offset = int(p[0]), int(p[1])
        return t


def key_table_utf8(keys: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    enc = [k.encode("utf-8") for k in keys]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    if enc:
        np.cumsum([len(e) for e in enc], out=off[1:])
    data = np.frombuffer(b"".join(enc), dtype=np.uint8).copy() if enc else np.zeros(0, np.uint8)
    return data, off


def encode_states(engine: ReplayEngine, template: JsonTemplate, d_keys_utf8, d_key_off, capacity_hint: int = 0,
                  envelope: str = "none"):
    """Encode every resident aggregate.  Returns ``(out, out_off)`` CUDA tensors: aggregate ``a``'s text is
    ``out[out_off[a]:out_off[a+1]]`` (empty for None / poisoned aggregates).

    ``envelope="protobuf_state"`` wraps each value in the multilanguage module's ``State{aggregateId, payload}``
    message (``multilanguage-protocol.proto:7-10``; what ``GenericSurgeCommandBusinessLogic.scala:36-39`` stores),
    with the template text as the payload."""
    import torch

    if envelope not in ("none", "protobuf_state"):
        raise ValueError(f"unknown envelope {envelope!r}")
    lib = _native.load()
    fn = lib.surge_replay_encode_json if envelope == "none" else lib.surge_replay_encode_protobuf_state
    n = engine.n_agg
    dev = d_key_off.device
    d_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    cap = int(capacity_hint) if capacity_hint else max(64, 2 * int(d_keys_utf8.numel()) + 64 * n)
    t = template.to_c()
    for _ in range(2):
        d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        total = ctypes.c_int64(0)
        rc = fn(
            engine._h, ctypes.byref(t), ctypes.c_void_p(d_keys_utf8.data_ptr()) if d_keys_utf8.numel() else None,
            ctypes.c_void_p(d_key_off.data_ptr()), ctypes.c_void_p(d_out.data_ptr()), cap,
            ctypes.c_void_p(d_off.data_ptr()), ctypes.byref(total))
        if rc == 0:
            return d_out[: total.value], d_off
        if rc == -6 and total.value > cap:  # SURGE_E_RANGE: retry with the exact size
            cap = total.value
            continue
        msg = lib.surge_replay_last_error(engine._h)
        raise ReplayError(rc, msg.decode() if msg else "")
    raise RuntimeError("encode_states: unreachable")
