// SOURCE-ONLY sketch (no JVM/sbt in the build image): how the engine plugs into an unmodified Surge
// node through the reference's own seams.  Python mirror with the same behaviour, exercised by the
// tests: surge_amd/store.py (GpuReplayStateStore, GpuReplayKeyValueStore, GpuReplayPersistencePlugin).
package surge.replay.gpu

import java.nio.{ ByteBuffer, ByteOrder }
import org.apache.kafka.common.utils.Bytes
import org.apache.kafka.streams.state.{ KeyValueBytesStoreSupplier, KeyValueStore }
import surge.kafka.streams.SurgeKafkaStreamsPersistencePlugin

/** What a replayable model declares beside its handleEvent (additive to AggregateCommandModel,
 *  CommandModels.scala:12-31): the event algebra and the fixed-width codecs. */
trait ReplayableModel[Agg, Evt] {
  def schema: ByteBuffer                                   // surge_replay_schema, little-endian
  def encodeEvent(evt: Evt, out: ByteBuffer): Unit         // 16 bytes: type, seq, payload
  def stateFromFixed(aggregateId: String, state64: ByteBuffer): Agg
}

/** Seam S1 — SurgeKafkaStreamsPersistencePlugin (SurgeKafkaStreamsPersistencePlugin.scala:12-15).
 *  Selected with:   surge.kafka-streams.state-store-plugin = "gpu-replay"
 *                   gpu-replay.plugin-class = "surge.replay.gpu.GpuReplayPersistencePlugin"
 *  (loader contract: SurgeKafkaStreamsPersistencePlugin.scala:30-50; needs a public no-arg constructor.)
 *  enableLogging = false: the store is rebuilt from the events topic and needs no changelog. */
class GpuReplayPersistencePlugin extends SurgeKafkaStreamsPersistencePlugin {
  override def enableLogging: Boolean = false
  override def createSupplier(storeName: String): KeyValueBytesStoreSupplier =
    new GpuReplayStoreSupplier(storeName, GpuReplayRegistry.recoveredFor(storeName))
}

/** KeyValueStore[Bytes, Array[Byte]] whose reads fall through to the GPU-recovered snapshot and whose
 *  puts (later state-topic records) overlay it: last write wins, null = tombstone
 *  (SurgeStateStoreConsumer.scala:69).  get() is what serves seam S2,
 *  AggregateStateStoreKafkaStreams.getAggregateBytes (AggregateStateStoreKafkaStreams.scala:83-85). */
final class GpuReplayKeyValueStore(name: String, recovered: RecoveredSnapshot) /* extends KeyValueStore[Bytes, Array[Byte]] */ {
  private val overlay = new java.util.concurrent.ConcurrentHashMap[String, Option[Array[Byte]]]()
  def put(key: Bytes, value: Array[Byte]): Unit = overlay.put(key.toString, Option(value))
  def get(key: Bytes): Array[Byte] = {
    val k = key.toString
    Option(overlay.get(k)) match {
      case Some(v) => v.orNull
      case None    => recovered.getAggregateBytes(k).orNull // surge_replay_get + the plugin's writeState
    }
  }
}

/** Recovery driver: pack the events topic into CSR (order by offset, group by aggregate id), fold on
 *  the GPU, publish the host mirror.  One instance per assigned state-topic partition / GPU. */
final class RecoveredSnapshot(handle: Long, keyIndex: java.util.Map[String, java.lang.Long], writeState: (String, ByteBuffer) => Array[Byte]) {
  def getAggregateBytes(aggregateId: String): Option[Array[Byte]] =
    Option(keyIndex.get(aggregateId)).flatMap { idx =>
      val st = ByteBuffer.allocateDirect(64).order(ByteOrder.LITTLE_ENDIAN)
      NativeReplay.get(handle, idx, st) match {
        case 1 => Some(writeState(aggregateId, st)) // bytes == aggregateWriteFormatting.writeState(state).value
        case 0 => None                               // KTable miss / tombstone
        case _ => throw new java.io.IOException("surge_replay_get failed") // => failed Future => fetchState retry
      }
    }
}
