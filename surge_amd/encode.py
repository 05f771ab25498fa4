"""GPU-side state encoders: fixed 64-byte states -> the plugin's serialized text, in bulk.

Point reads keep using the plugin's own ``writeState`` on the host (``surge_amd/store.py``); this is for
publishing a whole snapshot (10 M aggregates ≈ 1 GB of JSON) without a per-aggregate host loop.  The
text shape is declared as a template; ``JsonTemplate.counter()`` is the Counter fixture's play-json form
``{"aggregateId":"<id>","count":N,"version":N}``
(``modules/command-engine/scaladsl/src/test/scala/surge/scaladsl/TestBoundedContext.scala:15-16,127-129``);
``JsonTemplate.bank_account()`` is the surge-docs BankAccount's
``{"accountNumber":"<uuid>","accountOwner":"..","securityCode":"..","balance":<Double>}``
(``modules/surge-docs/src/test/scala/docs/command/BankAccountCommandModel.scala:19-23``, written by
``BankAccountSurgeModel.scala:26-28``): the Double as play-json writes it, the two strings from side columns.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Sequence, Tuple, Union

import numpy as np

from . import _native
from .replay import ReplayEngine, ReplayError

JP_LITERAL, JP_KEY, JP_I32, JP_U32, JP_I64, JP_F64, JP_STR = 0, 1, 2, 3, 4, 5, 6
STRING_COLUMNS = 4
MAX_PARTS = 16


class _CPart(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("field_offset", ctypes.c_uint32), ("lit_off", ctypes.c_uint32),
                ("lit_len", ctypes.c_uint32)]


class CJsonTemplate(ctypes.Structure):
    _fields_ = [("n_parts", ctypes.c_uint32), ("part", _CPart * MAX_PARTS), ("literals", ctypes.c_uint8 * 256)]


@dataclass(frozen=True)
class JsonTemplate:
    """Parts are ``bytes`` literals, the string ``"KEY"``, ``(kind, state_byte_offset)`` tuples, or ``(JP_STR, column)``."""

    parts: Sequence[Union[bytes, str, Tuple[int, int]]]

    @staticmethod
    def counter() -> "JsonTemplate":
        return JsonTemplate((b'{"aggregateId":', "KEY", b',"count":', (JP_I32, 0), b',"version":', (JP_I32, 4), b"}"))

    @staticmethod
    def bank_account() -> "JsonTemplate":
        """play-json's ``Json.format[BankAccount]``: fields in case-class order; ``balance`` is the v1 state's f64 at byte 16."""
        return JsonTemplate((b'{"accountNumber":', "KEY", b',"accountOwner":', (JP_STR, 0), b',"securityCode":', (JP_STR, 1),
                             b',"balance":', (JP_F64, 16), b"}"))

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


def play_json_double(x: float) -> str:
    """The text play-json writes for a Scala ``Double`` (``surge_format_f64_json``: the library's host copy of the
    conversion its GPU encoder uses).  Raises ``ValueError`` for NaN / infinities, as ``BigDecimal(NaN)`` throws."""
    buf = ctypes.create_string_buffer(32)
    n = _native.load().surge_format_f64_json(int(np.float64(x).view(np.uint64)), buf, 32)
    if n == 0:
        raise ValueError(f"{x!r} is not a JSON number (play-json: NumberFormatException)")
    return buf.raw[:n].decode("ascii")


def key_table_utf8(keys: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    enc = [k.encode("utf-8") for k in keys]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    if enc:
        np.cumsum([len(e) for e in enc], out=off[1:])
    data = np.frombuffer(b"".join(enc), dtype=np.uint8).copy() if enc else np.zeros(0, np.uint8)
    return data, off


def encode_states(engine: ReplayEngine, template: JsonTemplate, d_keys_utf8, d_key_off, capacity_hint: int = 0,
                  envelope: str = "none", strings=()):
    """Encode every resident aggregate.  Returns ``(out, out_off)`` CUDA tensors: aggregate ``a``'s text is
    ``out[out_off[a]:out_off[a+1]]`` (empty for None / poisoned aggregates).

    ``envelope="protobuf_state"`` wraps each value in the multilanguage module's ``State{aggregateId, payload}``
    message (``multilanguage-protocol.proto:7-10``; what ``GenericSurgeCommandBusinessLogic.scala:36-39`` stores),
    with the template text as the payload.  ``strings``: up to four ``(d_utf8, d_off)`` side string columns for
    ``(JP_STR, column)`` parts.  Raises ``ReplayError`` (UNSUPPORTED) after encoding everything else when some aggregate
    holds a NaN / infinite Double (no JSON number exists; those aggregates get zero bytes)."""
    import torch

    if envelope not in ("none", "protobuf_state"):
        raise ValueError(f"unknown envelope {envelope!r}")
    lib = _native.load()
    fn = lib.surge_replay_encode_json if envelope == "none" else lib.surge_replay_encode_protobuf_state
    n = engine.n_agg
    dev = d_key_off.device
    for c in range(STRING_COLUMNS):
        col = strings[c] if c < len(strings) else None
        engine._check(lib.surge_replay_set_encode_strings(
            engine._h, c, ctypes.c_void_p(col[0].data_ptr()) if col is not None and col[0].numel() else None,
            ctypes.c_void_p(col[1].data_ptr()) if col is not None else None))
    d_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    cap = int(capacity_hint) if capacity_hint else max(64, 2 * int(d_keys_utf8.numel()) + 64 * n + sum(2 * int(c[0].numel()) for c in strings))
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
