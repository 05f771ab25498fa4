// JNI binding of include/surge_replay.h; the native side is integration/jni/surge_replay_jni.c (every method below
// has its Java_surge_replay_gpu_NativeReplay_<name> export there, checked by tests/test_abi.py).  Source-only in this
// repository: the build image has no JDK / scalac; written against Scala 2.13 like the reference (build.sbt:6).
package surge.replay.gpu

import java.nio.ByteBuffer

/** All buffers are DIRECT ByteBuffers in native byte order; the shim checks address and capacity of every one. */
object NativeReplay {
  /** Loads libsurge_replay_jni.so (which links libsurge_replay.so, the HIP engine).  Failure is loud by design:
   *  the reference silently falls back to RocksDB when a plugin fails to load
   *  (SurgeKafkaStreamsPersistencePlugin.scala:34-47) — this plugin must never do that. */
  private val loaded: Boolean =
    try { System.loadLibrary("surge_replay_jni"); true }
    catch { case e: UnsatisfiedLinkError => throw new GpuReplayUnavailableException("libsurge_replay_jni.so / libsurge_replay.so not loadable", e) }

  def ensureLoaded(): Unit = require(loaded)

  @native def create(schema: ByteBuffer /* null = built-in algebra */, device: Int): Long
  @native def destroy(handle: Long): Unit
  @native def loadCsr(handle: Long, segOff: ByteBuffer, nAgg: Long, events: ByteBuffer, nEvents: Long, initState: ByteBuffer): Int
  @native def fold(handle: Long, algo: Int): Int
  /** Builds the index `algo` needs without folding (algo 7 = tile-major copy of the bound log); `out` (nullable, 24 bytes,
    * native order) receives index_build_ms, relayout_ms (doubles) and the copy's size in bytes (long). */
  @native def prepare(handle: Long, algo: Int, out: ByteBuffer): Int
  @native def appendFold(handle: Long, groupAgg: ByteBuffer, groupOff: ByteBuffer, nGroups: Long, events: ByteBuffer, nEvents: Long): Int
  @native def appendEvents(handle: Long, aggIdx: ByteBuffer, events: ByteBuffer, nEvents: Long): Int
  @native def grow(handle: Long, newNAgg: Long): Int
  @native def snapshot(handle: Long, nAgg: Long, states: ByteBuffer, present: ByteBuffer): Int
  /** 1 = Some (state64 filled), 0 = None, 2 = POISONED (replay hit a throwing event), -1 = error (IOException pending) */
  @native def get(handle: Long, aggIdx: Long, state64: ByteBuffer): Int
  @native def gather(handle: Long, aggIdx: ByteBuffer, n: Long, states: ByteBuffer): Int
  @native def partitionHash(utf16: ByteBuffer, strOff: ByteBuffer, n: Long, nPartitions: Int, partOut: ByteBuffer): Int
  @native def partitionHashUpToColon(utf16: ByteBuffer, strOff: ByteBuffer, n: Long, nPartitions: Int, partOut: ByteBuffer): Int
  @native def commUniqueId(idOut: ByteBuffer): Int
  @native def commInit(handle: Long, rank: Int, world: Int, id: ByteBuffer): Int
  @native def commDestroy(handle: Long): Int
  @native def commCounts(handle: Long, nLocal: Long, world: Int, countsOut: ByteBuffer): Long
  @native def allgatherSnapshot(handle: Long, nLocal: Long, slot: Int, mode: Int): Int
  @native def allgatherGroup(handles: ByteBuffer, n: Int, slot: Int): Int
  @native def gatheredRead(handle: Long, slot: Int, rank: Int, firstRow: Long, nRows: Long, states: ByteBuffer): Int

  // ---- device decode (include/surge_ingest.h): a poll's ConsumerRecords in bulk, no per-record JVM work ----
  /** template: a surge_event_json_template in native layout (how the plugin's JSON event text maps onto the 16-byte event),
    * or null when the record values already are 16-byte events. */
  @native def decoderCreate(template: ByteBuffer, device: Int): Long
  @native def decoderDestroy(decoder: Long): Unit
  /** keys / values: the records' bytes back to back; keyOff / valueOff: n + 1 offsets; offsets: n Kafka offsets (nullable). */
  @native def decoderPushRecords(decoder: Long, keys: ByteBuffer, keyOff: ByteBuffer, values: ByteBuffer, valueOff: ByteBuffer,
                                 offsets: ByteBuffer, n: Long): Int
  /** grow + device group-by + fold of everything pushed since the last call; out (nullable, 16 bytes): events folded, keys known. */
  @native def appendDecoded(handle: Long, decoder: Long, out: ByteBuffer): Int
  /** A recovery that folds once: keep this poll's decoded events on the device (topic order) instead of folding them ... */
  @native def stageDecoded(handle: Long, decoder: Long, out: ByteBuffer): Int
  /** ... and, at the topic's end, pack everything staged into one bound CSR log over nAgg aggregates: one fold(handle, algo) follows. */
  @native def packStaged(handle: Long, nAgg: Long): Int
  /** counts (16 bytes): keys, UTF-8 bytes; utf8Out / keyOffOut nullable for a size query. */
  @native def decoderKeys(decoder: Long, utf8Out: ByteBuffer, keyOffOut: ByteBuffer, counts: ByteBuffer): Int
}

final class GpuReplayUnavailableException(msg: String, cause: Throwable) extends RuntimeException(msg, cause)
