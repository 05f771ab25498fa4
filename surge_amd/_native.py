"""Builds and loads ``libsurge_replay.so`` (the C-ABI of ``include/surge_replay.h``) through ctypes.

There is no Python/CPU fallback for the fold: if the library cannot be built or loaded this
module raises, and every ``ReplayEngine`` call would fail loudly.
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
from typing import List, Optional

from .schema import CKernelInfo, CLayoutInfo, CSchema, CStats

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.abspath(os.path.join(_HERE, "..", "include"))
LIB_PATH = os.path.join(_HERE, "libsurge_replay.so")
def _read_sources() -> tuple:
    """The translation units, from ``csrc/SOURCES`` — the one list the Makefile reads too (tests/test_abi.py)."""
    with open(os.path.join(CSRC, "SOURCES")) as f:
        return tuple(ln.strip() for ln in f if ln.strip() and not ln.startswith("#"))


SOURCES = _read_sources()
# every header a translation unit may include or rtc.cpp embeds (.incbin): a change to any of them rebuilds every object
HEADERS = tuple(sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))) + tuple(
    os.path.join(INCLUDE, f) for f in ("surge_replay.h", "surge_ingest.h", "surge_snapshot.h"))

#: every symbol ``include/surge_replay.h`` declares (checked by tests/test_abi.py)
EXPORTS = (
    "surge_replay_default_schema",
    "surge_replay_create",
    "surge_replay_create_v2",
    "surge_replay_kernel_info",
    "surge_replay_compile_schema_v2",
    "surge_replay_compile_schema",
    "surge_replay_destroy",
    "surge_replay_last_error",
    "surge_replay_set_stream",
    "surge_replay_get_stream",
    "surge_replay_synchronize",
    "surge_replay_load_csr",
    "surge_replay_bind_device_csr",
    "surge_replay_fold",
    "surge_replay_prepare",
    "surge_replay_layout_info",
    "surge_replay_index_order",
    "surge_replay_stage_reserve",
    "surge_replay_stage_events_device",
    "surge_replay_staged",
    "surge_replay_pack_staged",
    "surge_replay_bound_log",
    "surge_replay_append_fold",
    "surge_replay_append_events",
    "surge_replay_append_events_device",
    "surge_replay_append_fold_device",
    "surge_replay_get",
    "surge_replay_gather",
    "surge_replay_snapshot",
    "surge_replay_device_state",
    "surge_replay_snapshot_delta",
    "surge_replay_snapshot_commit",
    "surge_replay_snapshot_invalidate",
    "surge_replay_set_encode_filter",
    "surge_replay_set_encode_strings",
    "surge_format_f64_json",
    "surge_format_f64_json_many",
    "surge_replay_encode_json",
    "surge_replay_encode_protobuf_state",
    "surge_replay_pack_states",
    "surge_replay_unpack_states",
    "surge_replay_partition_hash",
    "surge_replay_partition_hash_device",
    "surge_replay_partition_hash_up_to_colon",
    "surge_replay_partition_hash_up_to_colon_device",
    "surge_replay_set_state_out",
    "surge_replay_grow",
    "surge_replay_comm_unique_id",
    "surge_replay_comm_init",
    "surge_replay_comm_destroy",
    "surge_replay_comm_info",
    "surge_replay_comm_counts",
    "surge_replay_allgather_snapshot",
    "surge_replay_allgather",
    "surge_replay_comm_wait",
    "surge_replay_gathered",
    "surge_replay_gathered_read",
    "surge_replay_stats",
    "surge_replay_stats_reset",
    "surge_replay_fold_times",
    "surge_replay_stream_probe",
)

#: every symbol ``include/surge_ingest.h`` declares
INGEST_EXPORTS = (
    "surge_ingest_create",
    "surge_ingest_destroy",
    "surge_ingest_last_error",
    "surge_ingest_feed",
    "surge_ingest_set_threads",
    "surge_ingest_ready",
    "surge_ingest_drain",
    "surge_ingest_arena",
    "surge_ingest_drain_fixed16",
    "surge_ingest_drain_json",
    "surge_ingest_drain_sections",
    "surge_ingest_group_create",
    "surge_ingest_group_destroy",
    "surge_ingest_group_last_error",
    "surge_ingest_group_feed",
    "surge_ingest_group_receive_buffer",
    "surge_ingest_group_receive_copy",
    "surge_ingest_group_cpu_seconds",
    "surge_ingest_group_slab_bytes",
    "surge_ingest_group_queued_sections",
    "surge_ingest_group_counters",
    "surge_ingest_group_set_allocator",
    "surge_ingest_group_use_pinned_slabs",
    "surge_ingest_set_allocator",
    "surge_ingest_use_pinned_arena",
    "surge_device_decoder_create",
    "surge_device_decoder_destroy",
    "surge_device_decoder_last_error",
    "surge_device_decoder_push",
    "surge_device_decoder_push_records",
    "surge_device_decoder_push_async",
    "surge_device_decoder_push_parts_async",
    "surge_device_decoder_push_finish",
    "surge_device_decoder_push_finish_async",
    "surge_device_decoder_pending",
    "surge_device_decoder_reserve",
    "surge_device_decoder_result",
    "surge_device_decoder_clear",
    "surge_replay_append_decoded",
    "surge_replay_append_decoded_async",
    "surge_replay_stage_decoded",
    "surge_device_decoder_keys",
    "surge_device_decoder_key_table",
    "surge_device_decoder_counters",
    "surge_device_decoder_stats",
    "surge_event_json_validate",
    "surge_event_json_decode",
    "surge_event_json_last_error",
    "surge_parse_f64_json",
    "surge_ingest_key_count",
    "surge_ingest_key",
    "surge_ingest_counters",
    "surge_crc32c",
    "surge_crc32c_portable",
    "surge_lz4_frame_decompress",
    "surge_lz4_frame_bound",
    "surge_lz4_frame_compress",
    "surge_xxh32",
)

#: every symbol ``include/surge_snapshot.h`` declares
SNAPSHOT_EXPORTS = (
    "surge_snapshot_writer_create",
    "surge_snapshot_writer_destroy",
    "surge_snapshot_writer_set_compression",
    "surge_snapshot_writer_last_error",
    "surge_snapshot_writer_append",
    "surge_snapshot_writer_append_indexed",
    "surge_snapshot_writer_flush",
    "surge_snapshot_writer_partition",
    "surge_snapshot_writer_reset",
    "surge_device_framer_create",
    "surge_device_framer_destroy",
    "surge_device_framer_last_error",
    "surge_device_framer_frame",
    "surge_device_framer_next_offsets",
)

_lib: Optional[ctypes.CDLL] = None


class NativeLibraryError(RuntimeError):
    pass


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise NativeLibraryError("hipcc not found; cannot build libsurge_replay.so")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps: List[str] = [os.path.join(CSRC, s) for s in SOURCES] + list(HEADERS)
    return any(os.path.getmtime(d) > t for d in deps)


FLAGS = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall")
OBJ_DIR = os.path.join(_HERE, "build")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (the .so travels with the repo snapshot).

    One object per translation unit under ``surge_amd/build/`` (compiled side by side, only the ones older than their
    source or any header), then one link: the flags and the source list (``csrc/SOURCES``) are the Makefile's.
    """
    if not force and not needs_build():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor

    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in HEADERS)
    jobs = []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, src + ".o")
        path = os.path.join(CSRC, src)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(path), newest_header):
            jobs.append((path, obj))

    def compile_one(job):
        path, obj = job
        # -I csrc: rtc.cpp embeds the device headers with .incbin
        cmd = [hipcc, *FLAGS, "-I" + INCLUDE, "-I" + CSRC, "-c", path, "-o", obj + ".tmp"]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode == 0:
            os.replace(obj + ".tmp", obj)
        return path, proc

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
        for path, proc in pool.map(compile_one, jobs):
            if proc.returncode != 0:
                raise NativeLibraryError(f"hipcc failed on {path}:\n" + proc.stdout + proc.stderr)
            if verbose and (proc.stdout or proc.stderr):
                print(proc.stdout + proc.stderr)
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(OBJ_DIR, s + ".o") for s in SOURCES] + ["-o", tmp]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise NativeLibraryError("link failed:\n" + proc.stdout + proc.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


def load() -> ctypes.CDLL:
    """Load the library (building it first when the sources are newer).

    ``torch`` is imported first on purpose: its bundled ``libamdhip64.so`` has the same soname as
    the system one, and loading it first makes the library share torch's HIP runtime so device
    pointers and streams interoperate.
    """
    global _lib
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401  (side effect: loads torch's HIP runtime)
    except Exception:  # pragma: no cover - torch is optional for pure C hosts
        pass
    lib_path = os.environ.get("SURGE_REPLAY_LIB", LIB_PATH)  # experiments: load an alternative build
    if lib_path == LIB_PATH and needs_build():
        build()
    try:
        L = ctypes.CDLL(lib_path)
    except OSError as e:
        raise NativeLibraryError(f"cannot load {lib_path}: {e}") from e
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    sig = {
        "surge_replay_default_schema": ([ctypes.POINTER(CSchema)], i32),
        "surge_replay_create": ([ctypes.POINTER(CSchema), i32, ctypes.POINTER(vp)], i32),
        "surge_replay_create_v2": ([vp, i32, ctypes.POINTER(vp)], i32),
        "surge_replay_kernel_info": ([vp, ctypes.POINTER(CKernelInfo)], i32),
        "surge_replay_compile_schema_v2": ([vp, ctypes.c_char_p, vp, i64, ctypes.POINTER(i64)], i32),
        "surge_replay_compile_schema": ([vp, ctypes.c_char_p, vp, i64, ctypes.POINTER(i64)], i32),
        "surge_replay_destroy": ([vp], i32),
        "surge_replay_last_error": ([vp], ctypes.c_char_p),
        "surge_replay_set_stream": ([vp, vp], i32),
        "surge_replay_get_stream": ([vp, ctypes.POINTER(vp)], i32),
        "surge_replay_synchronize": ([vp], i32),
        "surge_replay_load_csr": ([vp, vp, i64, vp, i64, vp], i32),
        "surge_replay_bind_device_csr": ([vp, vp, i64, vp, i64, vp, vp], i32),
        "surge_replay_fold": ([vp, i32], i32),
        "surge_replay_prepare": ([vp, i32], i32),
        "surge_replay_layout_info": ([vp, ctypes.POINTER(CLayoutInfo)], i32),
        "surge_replay_index_order": ([vp, i32, vp, i64, ctypes.POINTER(i64)], i32),
        "surge_replay_stage_reserve": ([vp, i64], i32),
        "surge_replay_stage_events_device": ([vp, vp, vp, i64], i32),
        "surge_replay_staged": ([vp, ctypes.POINTER(i64)], i32),
        "surge_replay_pack_staged": ([vp, i64], i32),
        "surge_replay_bound_log": ([vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(i64)], i32),
        "surge_replay_append_fold": ([vp, vp, vp, i64, vp, i64], i32),
        "surge_replay_append_fold_device": ([vp, vp, vp, i64, vp, i64], i32),
        "surge_replay_append_events": ([vp, vp, vp, i64], i32),
        "surge_replay_append_events_device": ([vp, vp, vp, i64], i32),
        "surge_replay_get": ([vp, i64, vp, ctypes.POINTER(ctypes.c_uint8)], i32),
        "surge_replay_gather": ([vp, vp, i64, vp], i32),
        "surge_replay_snapshot": ([vp, vp, vp], i32),
        "surge_replay_device_state": ([vp, ctypes.POINTER(vp), ctypes.POINTER(i64)], i32),
        "surge_replay_snapshot_delta": ([vp, vp, ctypes.POINTER(i64), ctypes.POINTER(i64), i32], i32),
        "surge_replay_set_encode_filter": ([vp, vp], i32),
        "surge_replay_snapshot_commit": ([vp, vp], i32),
        "surge_replay_snapshot_invalidate": ([vp, vp], i32),
        "surge_replay_set_encode_strings": ([vp, i32, vp, vp], i32),
        "surge_format_f64_json": ([ctypes.c_uint64, vp, i32], i32),
        "surge_format_f64_json_many": ([vp, i64, vp, i64, vp], i64),
        "surge_replay_encode_json": ([vp, vp, vp, vp, vp, i64, vp, ctypes.POINTER(i64)], i32),
        "surge_replay_encode_protobuf_state": ([vp, vp, vp, vp, vp, i64, vp, ctypes.POINTER(i64)], i32),
        "surge_replay_pack_states": ([vp, vp, i64, vp, vp], i32),
        "surge_replay_unpack_states": ([vp, vp, i64, vp, vp], i32),
        "surge_replay_partition_hash": ([vp, vp, i64, i32, vp], i32),
        "surge_replay_partition_hash_device": ([vp, vp, vp, i64, i32, vp], i32),
        "surge_replay_partition_hash_up_to_colon": ([vp, vp, i64, i32, vp], i32),
        "surge_replay_partition_hash_up_to_colon_device": ([vp, vp, vp, i64, i32, vp], i32),
        "surge_replay_set_state_out": ([vp, vp], i32),
        "surge_replay_grow": ([vp, i64], i32),
        "surge_replay_comm_unique_id": ([vp], i32),
        "surge_replay_comm_init": ([vp, i32, i32, vp], i32),
        "surge_replay_comm_destroy": ([vp], i32),
        "surge_replay_comm_info": ([vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_char_p)], i32),
        "surge_replay_comm_counts": ([vp, i64, vp, ctypes.POINTER(i64)], i32),
        "surge_replay_allgather_snapshot": ([vp, vp, i64, vp, i64, i32, i32], i32),
        "surge_replay_allgather": ([vp, i32, vp, vp, i64, i32], i32),
        "surge_replay_comm_wait": ([vp, i32, i32], i32),
        "surge_replay_gathered": ([vp, i32, ctypes.POINTER(vp), ctypes.POINTER(i64)], i32),
        "surge_replay_gathered_read": ([vp, i32, i32, i64, i64, vp], i32),
        "surge_replay_stats": ([vp, ctypes.POINTER(CStats)], i32),
        "surge_replay_stats_reset": ([vp], i32),
        "surge_replay_fold_times": ([vp, vp, i64, ctypes.POINTER(i64)], i32),
        "surge_replay_stream_probe": ([vp, vp, i64, ctypes.POINTER(ctypes.c_double)], i32),
    }
    u8p = ctypes.POINTER(ctypes.c_uint8)
    sig.update({
        "surge_ingest_create": ([i32, ctypes.POINTER(vp)], i32),
        "surge_ingest_destroy": ([vp], i32),
        "surge_ingest_last_error": ([vp], ctypes.c_char_p),
        "surge_ingest_feed": ([vp, vp, i64, ctypes.POINTER(i64)], i32),
        "surge_ingest_set_threads": ([vp, i32], i32),
        "surge_ingest_ready": ([vp], i64),
        "surge_ingest_drain": ([vp, i64, vp, ctypes.POINTER(i64)], i32),
        "surge_ingest_arena": ([vp], vp),
        "surge_ingest_drain_fixed16": ([vp, i64, vp, vp, vp, ctypes.POINTER(i64)], i32),
        "surge_ingest_drain_json": ([vp, i64, vp, vp, vp, vp, ctypes.POINTER(i64)], i32),
        "surge_ingest_drain_sections": ([vp, i64, vp, ctypes.POINTER(i64)], i32),
        "surge_ingest_group_create": ([i32, i32, ctypes.POINTER(vp)], i32),
        "surge_ingest_group_destroy": ([vp], i32),
        "surge_ingest_group_last_error": ([vp], ctypes.c_char_p),
        "surge_ingest_group_feed": ([vp, vp, vp, i32, vp, i64, vp, ctypes.POINTER(i64), ctypes.POINTER(vp)], i32),
        "surge_ingest_group_receive_buffer": ([vp, i64, ctypes.POINTER(vp)], i32),
        "surge_ingest_group_receive_copy": ([vp, vp, vp, i32, vp], i32),
        "surge_ingest_group_cpu_seconds": ([vp, ctypes.POINTER(ctypes.c_double * 2)], i32),
        "surge_ingest_group_slab_bytes": ([vp, ctypes.POINTER(i64), ctypes.POINTER(i32)], i32),
        "surge_ingest_group_queued_sections": ([vp], i64),
        "surge_ingest_group_counters": ([vp, ctypes.POINTER(i64 * 8)], i32),
        "surge_ingest_group_set_allocator": ([vp, vp, vp], i32),
        "surge_ingest_group_use_pinned_slabs": ([vp], i32),
        "surge_ingest_set_allocator": ([vp, vp, vp], i32),
        "surge_ingest_use_pinned_arena": ([vp], i32),
        "surge_device_decoder_create": ([i32, vp, vp, ctypes.POINTER(vp)], i32),
        "surge_device_decoder_destroy": ([vp], i32),
        "surge_device_decoder_last_error": ([vp], ctypes.c_char_p),
        "surge_device_decoder_push": ([vp, vp, vp, i64], i32),
        "surge_device_decoder_push_records": ([vp, vp, vp, vp, vp, vp, i64], i32),
        "surge_device_decoder_push_async": ([vp, vp, vp, i64], i32),
        "surge_device_decoder_push_parts_async": ([vp, i32, vp, vp, vp], i32),
        "surge_device_decoder_push_finish": ([vp], i32),
        "surge_device_decoder_push_finish_async": ([vp], i32),
        "surge_device_decoder_pending": ([vp], i32),
        "surge_device_decoder_reserve": ([vp, i64, i64], i32),
        "surge_device_decoder_result": ([vp, ctypes.POINTER(i64), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i64)], i32),
        "surge_device_decoder_clear": ([vp], i32),
        "surge_replay_append_decoded": ([vp, vp, ctypes.POINTER(i64), ctypes.POINTER(i64)], i32),
        "surge_replay_append_decoded_async": ([vp, vp, ctypes.POINTER(i64), ctypes.POINTER(i64)], i32),
        "surge_replay_stage_decoded": ([vp, vp, ctypes.POINTER(i64), ctypes.POINTER(i64)], i32),
        "surge_device_decoder_keys": ([vp, vp, i64, vp, ctypes.POINTER(i64), ctypes.POINTER(i64)], i32),
        "surge_device_decoder_key_table": ([vp, ctypes.POINTER(vp), ctypes.POINTER(vp)], i32),
        "surge_device_decoder_counters": ([vp, ctypes.POINTER(i64 * 4)], i32),
        "surge_device_decoder_stats": ([vp, ctypes.POINTER(i64 * 8)], i32),
        "surge_event_json_validate": ([vp], i32),
        "surge_event_json_decode": ([vp, vp, i64, vp], i32),
        "surge_event_json_last_error": ([], ctypes.c_char_p),
        "surge_parse_f64_json": ([vp, i64, ctypes.POINTER(ctypes.c_uint64)], i32),
        "surge_ingest_key_count": ([vp], i64),
        "surge_ingest_key": ([vp, i64, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i64)], i32),
        "surge_ingest_counters": ([vp, ctypes.POINTER(i64 * 8)], i32),
        "surge_crc32c": ([vp, i64], ctypes.c_uint32),
        "surge_crc32c_portable": ([vp, i64], ctypes.c_uint32),
        "surge_lz4_frame_decompress": ([vp, i64, vp, i64], i64),
        "surge_lz4_frame_bound": ([i64], i64),
        "surge_lz4_frame_compress": ([vp, i64, vp, i64], i64),
        "surge_xxh32": ([vp, i64, ctypes.c_uint32], ctypes.c_uint32),
    })
    sig.update({
        "surge_snapshot_writer_create": ([i32, i32, i64, ctypes.POINTER(vp)], i32),
        "surge_snapshot_writer_destroy": ([vp], i32),
        "surge_snapshot_writer_last_error": ([vp], ctypes.c_char_p),
        "surge_snapshot_writer_append": ([vp, i64, vp, vp, vp, vp, vp, vp, i64], i32),
        "surge_snapshot_writer_append_indexed": ([vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, i64], i32),
        "surge_snapshot_writer_flush": ([vp], i32),
        "surge_snapshot_writer_partition": ([vp, i32, ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i64)], i32),
        "surge_snapshot_writer_reset": ([vp], i32),
        "surge_snapshot_writer_set_compression": ([vp, i32], i32),
        "surge_device_framer_create": ([i32, vp, i32, i32, i64, ctypes.POINTER(vp)], i32),
        "surge_device_framer_destroy": ([vp], i32),
        "surge_device_framer_last_error": ([vp], ctypes.c_char_p),
        "surge_device_framer_frame": ([vp, i64, vp, vp, vp, vp, vp, vp, i64, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(i64)], i32),
        "surge_device_framer_next_offsets": ([vp, vp], i32),
    })
    for name in EXPORTS + INGEST_EXPORTS + SNAPSHOT_EXPORTS:
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.argtypes, fn.restype = sig[name]
    _lib = L
    return L
