"""Synthetic event logs of the shapes BASELINE.json names, bit-identical on numpy and torch.

Every field of event ``i`` is a pure function of ``(seed, i, aggregate, position)`` through a
counter-based 32-bit integer hash, so the same log can be produced on the host (for the CPU
oracle), on the GPU (for HBM-resident benchmarks) and in arbitrary chunks, and they agree bit
for bit.  The reference has no fixed-width event format and no event reader
(``modules/serialization/src/main/scala/surge/core/SurgeFormatting.scala:9-11`` only writes), so
these logs stand in for a decoded events topic: per aggregate, events are contiguous and in publish
order — what one Kafka partition holds for keys ``"<id>:<seq>"``
(``.../scaladsl/src/test/scala/surge/scaladsl/TestBoundedContext.scala:122-124``).

Type mix for config C2: INC .45, DEC .35, NOOP .05, CREATE .05, SET_BALANCE .10; the
first event is forced to CREATE for half of the aggregates so both ``None`` paths are exercised.
``arg`` is uniform over all of int32 to exercise 32-bit wrap.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .schema import EVENT_DTYPE, EVT_CREATE, EVT_DEC, EVT_DELETE, EVT_INC, EVT_NOOP, EVT_SET_BALANCE, EVT_THROW

_M = 0xFFFFFFFF


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _hash32(x):
    """lowbias32 on int64 carriers holding values < 2**32 (works for numpy and torch)."""
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & _M
    x = ((x ^ (x >> 15)) * 0x846CA68B) & _M
    return x ^ (x >> 16)


def _h(seed: int, i, stream: int):
    k = (seed * 0x9E3779B1 + stream * 0x85EBCA77) & _M
    hi = _hash32(((i >> 32) + k) & _M)
    return _hash32((i & _M) ^ hi)


@dataclass(frozen=True)
class TypeMix:
    """Cumulative thresholds over [0, 2**32) for the event type draw."""

    inc: float = 0.45
    dec: float = 0.35
    noop: float = 0.05
    create: float = 0.05
    set_balance: float = 0.10
    delete: float = 0.0
    throw: float = 0.0
    force_create_half: bool = True

    def thresholds(self):
        probs = [
            (EVT_INC, self.inc),
            (EVT_DEC, self.dec),
            (EVT_NOOP, self.noop),
            (EVT_CREATE, self.create),
            (EVT_SET_BALANCE, self.set_balance),
            (EVT_DELETE, self.delete),
            (EVT_THROW, self.throw),
        ]
        total = sum(p for _, p in probs)
        acc, out = 0.0, []
        for t, p in probs:
            if p <= 0:
                continue
            acc += p / total
            out.append((t, min(int(acc * 2**32), 2**32)))
        out[-1] = (out[-1][0], 2**32)
        return out


C2_MIX = TypeMix()
#: the Counter fixture only (config C1): uniform over {INC, DEC, NOOP}, args in [1, 9]
C1_MIX = TypeMix(inc=1 / 3, dec=1 / 3, noop=1 / 3, create=0, set_balance=0, force_create_half=False)
#: everything, including tombstones and throwing events (parity stress)
STRESS_MIX = TypeMix(inc=0.40, dec=0.30, noop=0.05, create=0.06, set_balance=0.12, delete=0.05, throw=0.02)


def event_words(idx, agg, pos, seed: int, mix: TypeMix = C2_MIX, small_args: bool = False):
    """Events ``idx`` (int64 global indices) as an ``[n, 2]`` int64 array of little-endian words.

    ``agg``/``pos`` are each event's aggregate index and position inside its segment.
    word0 = type | seq << 32 ; word1 = payload (``arg`` zero-extended, or the f64 bits of ``value``).
    """
    torch_mode = _is_torch(idx)
    if torch_mode:
        import torch

        where, stack = torch.where, lambda a, b: torch.stack((a, b), dim=1)
        const = lambda v: torch.full_like(idx, v)  # noqa: E731
    else:
        where, stack = np.where, lambda a, b: np.stack((a, b), axis=1)
        const = lambda v: np.full_like(idx, v)  # noqa: E731

    r1 = _h(seed, idx, 1)
    r2 = _h(seed, idx, 2)
    ty = const(mix.thresholds()[-1][0])
    for t, thr in reversed(mix.thresholds()[:-1]):
        ty = where(r1 < thr, const(t), ty)
    if mix.force_create_half:
        forced = (pos == 0) & ((_h(seed, agg, 7) & 1) == 1)
        ty = where(forced, const(EVT_CREATE), ty)

    # integer payload: full-range int32 (wraps), or 1..9 for the C1 plumbing config
    arg = (r2 % 9 + 1) if small_args else r2
    # f64 payload: exactly representable cents-like value in [-65536, 65536)
    cents = (r2 & 0xFFFFFF) - 0x800000
    if torch_mode:
        import torch

        val_bits = (cents.to(torch.float64) / 128.0).view(torch.int64)
    else:
        val_bits = (cents.astype(np.float64) / 128.0).view(np.int64)
    is_f64 = (ty == EVT_CREATE) | (ty == EVT_SET_BALANCE)
    payload = where(is_f64, val_bits, arg)

    seq = (pos + 1) & 0x7FFFFFFF
    word0 = ty | (seq << 32)
    return stack(word0, payload)


def zipf_cdf(max_len: int = 4096) -> np.ndarray:
    """CDF of P(k) ∝ 1/k on k = 1..max_len."""
    w = 1.0 / np.arange(1, max_len + 1, dtype=np.float64)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    cdf[-1] = 1.0
    return cdf


def zipf_lengths(agg, seed: int, max_len: int = 4096):
    """Events per aggregate for aggregate indices ``agg`` (int64; numpy or torch)."""
    cdf = zipf_cdf(max_len)
    r = _h(seed, agg, 3)
    if _is_torch(agg):
        import torch

        u = (r.to(torch.float64) + 0.5) / 4294967296.0
        k = torch.searchsorted(torch.from_numpy(cdf).to(agg.device), u, right=False)
        return torch.clamp(k, max=max_len - 1) + 1
    u = (r.astype(np.float64) + 0.5) / 4294967296.0
    k = np.searchsorted(cdf, u, side="left")
    return np.minimum(k, max_len - 1).astype(np.int64) + 1


def fixed_log(n_agg: int, events_per_agg: int, seed: int, mix: TypeMix = C2_MIX, small_args: bool = False):
    """Host (numpy) log with a fixed fan-in: returns ``(seg_off, events)``."""
    n = n_agg * events_per_agg
    idx = np.arange(n, dtype=np.int64)
    agg = idx // max(events_per_agg, 1)
    pos = idx - agg * events_per_agg
    words = event_words(idx, agg, pos, seed, mix, small_args)
    seg_off = np.arange(n_agg + 1, dtype=np.int64) * events_per_agg
    return seg_off, np.ascontiguousarray(words).view(EVENT_DTYPE).reshape(-1)


def csr_log(lengths: np.ndarray, seed: int, mix: TypeMix = C2_MIX, small_args: bool = False):
    """Host (numpy) log with the given per-aggregate event counts."""
    lengths = np.asarray(lengths, dtype=np.int64)
    seg_off = np.zeros(lengths.shape[0] + 1, dtype=np.int64)
    np.cumsum(lengths, out=seg_off[1:])
    n = int(seg_off[-1])
    idx = np.arange(n, dtype=np.int64)
    agg = np.searchsorted(seg_off, idx, side="right") - 1
    pos = idx - seg_off[agg]
    words = event_words(idx, agg, pos, seed, mix, small_args)
    return seg_off, np.ascontiguousarray(words).view(EVENT_DTYPE).reshape(-1)


def zipf_log(n_agg: int, seed: int, max_len: int = 4096, mix: TypeMix = C2_MIX):
    return csr_log(zipf_lengths(np.arange(n_agg, dtype=np.int64), seed, max_len), seed, mix)


# ---- device-side generation (torch), chunked so multi-GB logs never need a host copy -------------

def fixed_log_device(n_agg: int, events_per_agg: int, seed: int, device, mix: TypeMix = C2_MIX,
                     chunk_events: int = 1 << 25, first_agg: int = 0):
    """``(seg_off, events[n, 2] int64)`` on ``device``; aggregates are ``first_agg .. first_agg+n_agg``
    of the global log (so shards of one log can be generated independently per rank)."""
    import torch

    n = n_agg * events_per_agg
    events = torch.empty((n, 2), dtype=torch.int64, device=device)
    base = first_agg * events_per_agg
    for s in range(0, n, chunk_events):
        e = min(n, s + chunk_events)
        idx = torch.arange(base + s, base + e, dtype=torch.int64, device=device)
        agg = idx // events_per_agg
        pos = idx - agg * events_per_agg
        events[s:e] = event_words(idx, agg, pos, seed, mix)
    seg_off = torch.arange(n_agg + 1, dtype=torch.int64, device=device) * events_per_agg
    return seg_off, events


def csr_log_device(lengths, seed: int, mix: TypeMix = C2_MIX, chunk_events: int = 1 << 25,
                   agg_ids=None, global_seg_off=None):
    """Device log for per-aggregate counts ``lengths`` (int64 CUDA tensor).

    With ``agg_ids``/``global_seg_off`` the events are those of the listed aggregates of a larger
    global log (their global event indices feed the hash), which is how a shard is generated.
    """
    import torch

    device = lengths.device
    seg_off = torch.zeros(lengths.numel() + 1, dtype=torch.int64, device=device)
    torch.cumsum(lengths, 0, out=seg_off[1:])
    n = int(seg_off[-1].item())
    events = torch.empty((n, 2), dtype=torch.int64, device=device)
    for s in range(0, n, chunk_events):
        e = min(n, s + chunk_events)
        local = torch.arange(s, e, dtype=torch.int64, device=device)
        a_local = torch.searchsorted(seg_off, local, right=True) - 1
        pos = local - seg_off[a_local]
        if agg_ids is None:
            idx, agg = local, a_local
        else:
            agg = agg_ids[a_local]
            idx = global_seg_off[agg] + pos
        events[s:e] = event_words(idx, agg, pos, seed, mix)
    return seg_off, events


def to_event_records(words) -> np.ndarray:
    """``[n, 2]`` int64 words (numpy or CPU/GPU torch) -> numpy ``EVENT_DTYPE`` records."""
    if _is_torch(words):
        words = words.detach().cpu().numpy()
    return np.ascontiguousarray(words).view(EVENT_DTYPE).reshape(-1)


def fixed_log_for_aggregates_device(agg_ids, events_per_agg: int, seed: int, mix: TypeMix = C2_MIX,
                                    chunk_aggs: int = 1 << 17):
    """Shard of the global fixed-fan-in log: the events of the listed global aggregate indices
    (int64 tensor, any device), in list order.  Returns ``(seg_off, events[n, 2])``."""
    import torch

    device = agg_ids.device
    n_agg = agg_ids.numel()
    L = events_per_agg
    events = torch.empty((n_agg * L, 2), dtype=torch.int64, device=device)
    pos1 = torch.arange(L, dtype=torch.int64, device=device)
    for s in range(0, n_agg, chunk_aggs):
        e = min(n_agg, s + chunk_aggs)
        agg = agg_ids[s:e].repeat_interleave(L)
        pos = pos1.repeat(e - s)
        idx = agg * L + pos
        events[s * L: e * L] = event_words(idx, agg, pos, seed, mix)
    seg_off = torch.arange(n_agg + 1, dtype=torch.int64, device=device) * L
    return seg_off, events
