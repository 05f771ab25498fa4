"""The host-side decoders take untrusted bytes (a Kafka fetch): mutation-fuzz them under AddressSanitizer + UBSan.
Builds ingest.cpp / event_decode.cpp / f64_text.cpp / lz4_frame.cpp / snapshot_writer.cpp with g++ -fsanitize=address,undefined (plain C++: no HIP in those
files) and drives the result from a subprocess that preloads libasan."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _asan_runtime():
    try:
        p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.parametrize("seed", [1, 2])
def test_mutated_record_batches_and_lz4_frames_never_touch_memory_they_do_not_own(tmp_path, seed):
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("no libasan.so next to gcc")
    lib = str(tmp_path / "libsurge_host_asan.so")
    srcs = [os.path.join(ROOT, "surge_amd", "csrc", f) for f in ("ingest.cpp", "event_decode.cpp", "f64_text.cpp", "lz4_frame.cpp", "snapshot_writer.cpp")]
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                    "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include")] + srcs + ["-o", lib, "-lpthread"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    res = subprocess.run([sys.executable, os.path.join(HERE, "fuzz_host_worker.py"), lib, "6", str(seed)], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0 and "OK " in res.stdout, (res.stdout[-1500:] + res.stderr[-6000:])


def test_multithreaded_snapshot_writer_is_race_free_under_thread_sanitizer(tmp_path):
    """The writer frames partitions on up to 16 host threads (plain and LZ4): tests/cpp/tsan_snapshot_writer.cpp appends
    2 x 200 k records over 64 partitions under -fsanitize=thread and decodes every partition again with the product's reader."""
    exe = str(tmp_path / "tsan_snapshot_writer")
    srcs = [os.path.join(HERE, "cpp", "tsan_snapshot_writer.cpp")] + [os.path.join(ROOT, "surge_amd", "csrc", f) for f in ("snapshot_writer.cpp", "lz4_frame.cpp", "ingest.cpp", "event_decode.cpp", "f64_text.cpp")]
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include")] + srcs + ["-lpthread", "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and "tsan" in (build.stderr or "").lower():
        pytest.skip("no ThreadSanitizer runtime next to g++")
    assert build.returncode == 0, build.stderr[-3000:]
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert res.returncode == 0 and res.stdout.count("PASS") == 2 and "ThreadSanitizer" not in res.stderr, res.stdout + res.stderr[-4000:]


def test_ingest_group_pool_slabs_and_undo_are_race_free_under_thread_sanitizer(tmp_path):
    """surge_ingest_group_feed on 8 pool threads through more than two turns of the six slabs while a consumer thread reads
    the slabs of up to five earlier fetches (what device pushes in flight do), with failing feeds undone in between
    (tests/cpp/tsan_ingest_group.cpp): no report from -fsanitize=thread, and every fetch's sections equal what each
    partition's own framer delivers."""
    exe = str(tmp_path / "tsan_ingest_group")
    srcs = [os.path.join(HERE, "cpp", "tsan_ingest_group.cpp")] + [os.path.join(ROOT, "surge_amd", "csrc", f) for f in ("ingest.cpp", "event_decode.cpp", "f64_text.cpp", "lz4_frame.cpp")]
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include")] + srcs + ["-lpthread", "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and "tsan" in (build.stderr or "").lower():
        pytest.skip("no ThreadSanitizer runtime next to g++")
    assert build.returncode == 0, build.stderr[-3000:]
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert res.returncode == 0 and "PASS group" in res.stdout and "ThreadSanitizer" not in res.stderr, res.stdout + res.stderr[-4000:]
