// How the engine plugs into an unmodified Surge node through the reference's own seams (SURVEY §8b S1 / S2).
// Source-only in this repository (no JDK / scalac / Kafka jars in the build image); complete against
// kafka-streams 3.2.3 (Dependencies.scala:42) — every abstract member of KeyValueStore[Bytes, Array[Byte]] and
// KeyValueBytesStoreSupplier is implemented, the same set the reference's own test store implements
// (modules/common/src/test/scala/surge/kafka/streams/SingleExceptionThrowingKeyValueStore.scala).
// Behavioural twin, exercised by the test-suite on the GPU: surge_amd/store.py.
package surge.replay.gpu

import java.nio.{ ByteBuffer, ByteOrder }
import java.util
import java.util.concurrent.{ ConcurrentHashMap, ConcurrentSkipListMap }

import org.apache.kafka.common.utils.Bytes
import org.apache.kafka.streams.KeyValue
import org.apache.kafka.streams.processor.{ ProcessorContext, StateStore }
import org.apache.kafka.streams.state.{ KeyValueBytesStoreSupplier, KeyValueIterator, KeyValueStore }
import surge.kafka.streams.SurgeKafkaStreamsPersistencePlugin

import scala.jdk.CollectionConverters._

/** What a replayable model declares beside its handleEvent (additive to AggregateCommandModel,
 *  CommandModels.scala:12-31): the event algebra and the fixed-width codecs. */
trait ReplayableModel[Agg, Evt] {
  /** surge_replay_schema (include/surge_replay.h), little-endian, direct buffer */
  def schema: ByteBuffer
  /** 16 bytes at out.position(): type, sequenceNumber, payload */
  def encodeEvent(evt: Evt, out: ByteBuffer): Unit
  def aggregateIdOf(evt: Evt): String
  /** bytes == aggregateWriteFormatting.writeState(state).value for the state held in the 64-byte record */
  def writeStateFromFixed(aggregateId: String, state64: ByteBuffer): Array[Byte]
}

/** Seam S1 — SurgeKafkaStreamsPersistencePlugin (SurgeKafkaStreamsPersistencePlugin.scala:12-15).
 *  Selected with:   surge.kafka-streams.state-store-plugin = "gpu-replay"
 *                   gpu-replay.plugin-class = "surge.replay.gpu.GpuReplayPersistencePlugin"
 *  (loader contract: SurgeKafkaStreamsPersistencePlugin.scala:30-50 — public no-arg constructor; ANY failure in it makes
 *  the loader fall back to RocksDB with only a log line, :34-47, so the constructor does nothing that can fail and
 *  createSupplier is where a missing native library / missing recovery surfaces, loudly.)
 *  enableLogging = false: the store is rebuilt from the events topic and needs no changelog; the reference then builds
 *  the topology un-optimised (SurgeStateStoreConsumer.scala:63-75). */
class GpuReplayPersistencePlugin extends SurgeKafkaStreamsPersistencePlugin {
  override def enableLogging: Boolean = false
  override def createSupplier(storeName: String): KeyValueBytesStoreSupplier = {
    NativeReplay.ensureLoaded() // throws GpuReplayUnavailableException: no silent RocksDB fallback (SURVEY appendix B)
    new GpuReplayStoreSupplier(storeName, GpuReplayRegistry.recoveredFor(storeName))
  }
}

final class GpuReplayStoreSupplier(val name: String, recovered: RecoveredSnapshot) extends KeyValueBytesStoreSupplier {
  override def get(): KeyValueStore[Bytes, Array[Byte]] = new GpuReplayKeyValueStore(name, recovered)
  override def metricsScope(): String = "gpu-replay"
}

/** storeName ("<aggregateName>AggregateStateStore", SurgeStateStoreConsumer.scala:110) -> the recovery that backs it.
 *  The application registers a recovery before it starts the Surge engine; a store created without one is a
 *  configuration error and fails the stream thread (never an empty store that looks like "no aggregates"). */
object GpuReplayRegistry {
  private val recoveries = new ConcurrentHashMap[String, RecoveredSnapshot]()
  def register(storeName: String, recovered: RecoveredSnapshot): Unit = recoveries.put(storeName, recovered)
  def unregister(storeName: String): Unit = Option(recoveries.remove(storeName)).foreach(_.close())
  def recoveredFor(storeName: String): RecoveredSnapshot =
    Option(recoveries.get(storeName)).getOrElse(throw new IllegalStateException(
      s"no GPU recovery registered for state store [$storeName]: call GpuReplayRegistry.register before starting the engine"))
}

/** KeyValueStore[Bytes, Array[Byte]] whose reads fall through to the GPU-recovered snapshot and whose puts (the
 *  state-topic records the KTable topology keeps indexing after recovery) overlay it: last write wins, null =
 *  tombstone (SurgeStateStoreConsumer.scala:69).  get() is what serves seam S2,
 *  AggregateStateStoreKafkaStreams.getAggregateBytes (AggregateStateStoreKafkaStreams.scala:83-85), on the 32-thread
 *  IO pool (ThreadPools.scala:10-11): the overlay is concurrent, the native get is lock-free against the mirror. */
final class GpuReplayKeyValueStore(val name: String, recovered: RecoveredSnapshot) extends KeyValueStore[Bytes, Array[Byte]] {
  private val Tombstone = new Array[Byte](0)
  private val overlay = new ConcurrentSkipListMap[Bytes, Array[Byte]]() // Bytes orders lexicographically, like RocksDB
  @volatile private var open = false

  private def key(k: Bytes): String = new String(k.get(), java.nio.charset.StandardCharsets.UTF_8)

  override def get(k: Bytes): Array[Byte] = {
    val o = overlay.get(k)
    if (o ne null) { if (o eq Tombstone) null else o }
    else recovered.getAggregateBytes(key(k)).orNull // surge_replay_get + the plugin's writeState
  }
  override def put(k: Bytes, value: Array[Byte]): Unit = overlay.put(k, if (value eq null) Tombstone else value)
  override def putIfAbsent(k: Bytes, value: Array[Byte]): Array[Byte] = {
    val prev = get(k)
    if (prev eq null) put(k, value)
    prev
  }
  override def putAll(entries: util.List[KeyValue[Bytes, Array[Byte]]]): Unit = entries.asScala.foreach(kv => put(kv.key, kv.value))
  override def delete(k: Bytes): Array[Byte] = {
    val prev = get(k)
    put(k, null)
    prev
  }

  /** key-ordered view of overlay ∪ recovered ids, tombstones removed (KafkaStreamsKeyValueStoreSpec.scala:37-91) */
  private def view(from: Option[Bytes], to: Option[Bytes]): KeyValueIterator[Bytes, Array[Byte]] = {
    val keys = new util.TreeSet[Bytes](overlay.keySet())
    recovered.aggregateIds.foreach(id => keys.add(Bytes.wrap(id.getBytes(java.nio.charset.StandardCharsets.UTF_8))))
    val in = keys.asScala.iterator
      .filter(k => from.forall(k.compareTo(_) >= 0) && to.forall(k.compareTo(_) <= 0)) // range is inclusive on both ends
      .flatMap(k => Option(get(k)).map(v => new KeyValue[Bytes, Array[Byte]](k, v)))
      .buffered
    new KeyValueIterator[Bytes, Array[Byte]] {
      override def close(): Unit = ()
      override def peekNextKey(): Bytes = in.head.key
      override def hasNext: Boolean = in.hasNext
      override def next(): KeyValue[Bytes, Array[Byte]] = in.next()
    }
  }
  override def range(from: Bytes, to: Bytes): KeyValueIterator[Bytes, Array[Byte]] = view(Option(from), Option(to))
  override def all(): KeyValueIterator[Bytes, Array[Byte]] = view(None, None)
  override def approximateNumEntries(): Long = recovered.aggregateIds.size.toLong + overlay.size()

  override def init(context: ProcessorContext, root: StateStore): Unit = {
    // logging is disabled for this store, so the restore callback is never fed a changelog; it exists because
    // register() is how Kafka Streams learns the store is initialised
    context.register(root, (k: Array[Byte], v: Array[Byte]) => put(Bytes.wrap(k), v))
    open = true
  }
  override def flush(): Unit = ()
  override def close(): Unit = open = false
  override def persistent(): Boolean = false // nothing on local disk: recovered from the events topic every time
  override def isOpen: Boolean = open
}

/** One recovered shard: the native handle (one GPU / one assigned state-topic partition set), the key table
 *  (aggregate id -> dense index; ids never cross the C ABI) and the model's fixed-width -> bytes codec. */
final class RecoveredSnapshot(private[gpu] val handle: Long, keyIndex: util.Map[String, java.lang.Long], model: ReplayableModel[_, _]) extends AutoCloseable {
  private val scratch = ThreadLocal.withInitial[ByteBuffer](() => ByteBuffer.allocateDirect(64).order(ByteOrder.LITTLE_ENDIAN))

  def aggregateIds: Iterable[String] = keyIndex.keySet().asScala

  /** Some(bytes) / None exactly like getAggregateBytes; a failed read or a POISONED aggregate throws, which fails the
   *  Future and sends KTableInitializationSupport.fetchState into its retry loop (:63-81), ending — after
   *  max-initialization-attempts — in ACKError(AggregateInitializationException) (PersistentActor.scala:328-333). */
  def getAggregateBytes(aggregateId: String): Option[Array[Byte]] =
    Option(keyIndex.get(aggregateId)).flatMap { idx =>
      val st = scratch.get()
      st.clear()
      NativeReplay.get(handle, idx, st) match {
        case 1 => Some(model.writeStateFromFixed(aggregateId, st))
        case 0 => None // KTable miss / tombstone
        case 2 => throw new GpuReplayPoisonedAggregateException(aggregateId)
        case _ => throw new java.io.IOException(s"surge_replay_get failed for $aggregateId")
      }
    }

  override def close(): Unit = NativeReplay.destroy(handle)
}

final class GpuReplayPoisonedAggregateException(aggregateId: String)
    extends RuntimeException(s"replay of aggregate [$aggregateId] hit an event whose handler throws; its state is frozen before that event and is not served")

/** Recovery driver: pack the events topic of the partitions this node owns into CSR (order by offset, group by
 *  aggregate id = record key up to ':', PartitionStringUpToColon), fold on the GPU, publish the host mirror. */
object GpuReplayRecovery {
  /** records: (key, value-as-16-byte-fixed-event) in offset order, already filtered to read_committed
   *  (SurgeStateStoreConsumer.scala:38); the native ingest (include/surge_ingest.h) produces exactly this shape. */
  def recover[Agg, Evt](model: ReplayableModel[Agg, Evt], device: Int, records: Iterator[(String, Array[Byte])]): RecoveredSnapshot = {
    NativeReplay.ensureLoaded()
    val keyIndex = new util.HashMap[String, java.lang.Long]()
    val perAgg = new util.ArrayList[util.ArrayList[Array[Byte]]]()
    records.foreach { case (key, ev) =>
      val id = key.takeWhile(_ != ':')
      val idx = keyIndex.computeIfAbsent(id, _ => { perAgg.add(new util.ArrayList[Array[Byte]]()); java.lang.Long.valueOf(perAgg.size() - 1L) })
      perAgg.get(idx.intValue()).add(ev)
    }
    val nAgg = perAgg.size()
    val nEvents = perAgg.asScala.map(_.size().toLong).sum
    val segOff = ByteBuffer.allocateDirect((nAgg + 1) * 8).order(ByteOrder.LITTLE_ENDIAN)
    val events = ByteBuffer.allocateDirect(math.max(16L, nEvents * 16).toInt).order(ByteOrder.LITTLE_ENDIAN)
    var off = 0L
    segOff.putLong(0L)
    perAgg.asScala.foreach { evs => evs.asScala.foreach(e => events.put(e, 0, 16)); off += evs.size(); segOff.putLong(off) }
    val h = NativeReplay.create(model.schema, device) // IOException("... no CPU fallback") without a GPU
    NativeReplay.loadCsr(h, segOff, nAgg.toLong, events, nEvents, null)
    NativeReplay.fold(h, 0)
    NativeReplay.snapshot(h, nAgg.toLong, null, null) // publishes the mirror that serves the 32 concurrent readers
    new RecoveredSnapshot(h, keyIndex, model)
  }
}

/** The same recovery with NO per-record JVM work: every poll of the events-topic consumer (key / value byte arrays, as a
 *  ByteArrayDeserializer hands them over, read_committed) is copied into direct buffers and pushed to a device decoder —
 *  key parsing, aggregate-id interning, event decoding (16-byte events, or the plugin's play-json text through
 *  `eventTemplate`: include/surge_ingest.h surge_event_json_template) and the group-by + fold all run on the GPU; the JVM
 *  crosses JNI twice per poll.  Replaces the HashMap / ArrayList loop of `recover` above for topics of any size. */
object GpuReplayBulkRecovery {
  /** foldOnce = true: a recovery that folds the topic ONCE — every poll's decoded events are staged on the device
   *  (stageDecoded) instead of folded, finish() packs them into one bound CSR log (packStaged) and runs a single fold of the
   *  kernel the library picks for that log (the lane-per-row kernels for a log large enough); the log stays bound. */
  final class Session(model: ReplayableModel[_, _], device: Int, eventTemplate: ByteBuffer, foldOnce: Boolean = false) extends AutoCloseable {
    NativeReplay.ensureLoaded()
    private val handle = NativeReplay.create(model.schema, device)
    private val decoder = NativeReplay.decoderCreate(eventTemplate, device)
    locally { // an empty resident state to grow: zero aggregates, folded
      val segOff = ByteBuffer.allocateDirect(8).order(ByteOrder.LITTLE_ENDIAN)
      segOff.putLong(0, 0L)
      NativeReplay.loadCsr(handle, segOff, 0L, null, 0L, null)
      NativeReplay.fold(handle, 0)
    }

    /** One poll: records in offset order per partition. */
    def push(records: IndexedSeq[(Array[Byte], Array[Byte], Long)]): Unit = if (records.nonEmpty) {
      val n = records.size
      val keyBytes = records.iterator.map(r => if (r._1 eq null) 0L else r._1.length.toLong).sum
      val valBytes = records.iterator.map(r => if (r._2 eq null) 0L else r._2.length.toLong).sum
      val keys = ByteBuffer.allocateDirect(math.max(1L, keyBytes).toInt)
      val values = ByteBuffer.allocateDirect(math.max(1L, valBytes).toInt)
      val keyOff = ByteBuffer.allocateDirect((n + 1) * 8).order(ByteOrder.LITTLE_ENDIAN)
      val valOff = ByteBuffer.allocateDirect((n + 1) * 8).order(ByteOrder.LITTLE_ENDIAN)
      val offsets = ByteBuffer.allocateDirect(n * 8).order(ByteOrder.LITTLE_ENDIAN)
      keyOff.putLong(0L); valOff.putLong(0L)
      records.foreach { case (k, v, o) =>
        if (k ne null) keys.put(k)
        if (v ne null) values.put(v)
        keyOff.putLong(keys.position().toLong); valOff.putLong(values.position().toLong); offsets.putLong(o)
      }
      NativeReplay.decoderPushRecords(decoder, keys, keyOff, values, valOff, offsets, n.toLong)
      if (foldOnce) NativeReplay.stageDecoded(handle, decoder, null) else NativeReplay.appendDecoded(handle, decoder, null)
    }

    /** End of the topic: publish the host mirror that serves the 32 concurrent readers and hand over the store. */
    def finish(): RecoveredSnapshot = {
      val counts = ByteBuffer.allocateDirect(16).order(ByteOrder.LITTLE_ENDIAN)
      NativeReplay.decoderKeys(decoder, null, null, counts)
      val nKeys = counts.getLong(0)
      if (foldOnce) { // everything staged -> one bound log -> one fold
        NativeReplay.packStaged(handle, nKeys)
        NativeReplay.fold(handle, 0)
      }
      val utf8 = ByteBuffer.allocateDirect(math.max(1L, counts.getLong(8)).toInt)
      val keyOff = ByteBuffer.allocateDirect(((nKeys + 1) * 8).toInt).order(ByteOrder.LITTLE_ENDIAN)
      NativeReplay.decoderKeys(decoder, utf8, keyOff, counts)
      val keyIndex = new util.HashMap[String, java.lang.Long]()
      var i = 0L
      while (i < nKeys) {
        val a = keyOff.getLong((i * 8).toInt).toInt
        val b = keyOff.getLong(((i + 1) * 8).toInt).toInt
        val bytes = new Array[Byte](b - a)
        utf8.position(a); utf8.get(bytes)
        keyIndex.put(new String(bytes, java.nio.charset.StandardCharsets.UTF_8), java.lang.Long.valueOf(i))
        i += 1
      }
      NativeReplay.snapshot(handle, nKeys, null, null)
      NativeReplay.decoderDestroy(decoder)
      new RecoveredSnapshot(handle, keyIndex, model)
    }

    override def close(): Unit = { NativeReplay.decoderDestroy(decoder); NativeReplay.destroy(handle) }
  }
}

/** One JVM that owns every GPU of the node (one RecoveredSnapshot per device): after the shards are folded, every device
 *  gets every shard's final states — the path's single exchange step (SURVEY §8e) — without RCCL or a rendezvous:
 *  surge_replay_allgather moves the shards with peer copies over xGMI.  Shard r = the state-topic partitions p with
 *  p % shards.size == r (PartitionAssignments.scala:51-63 gives the node its partitions; KafkaPartitioner.scala:8 the
 *  partition of a key). */
object GpuReplayNodeExchange {
  /** Asynchronous on every handle's side stream; `slot` (0 / 1) lets the next fold overlap it. */
  def exchange(shards: IndexedSeq[RecoveredSnapshot], slot: Int): Unit = {
    val hs = ByteBuffer.allocateDirect(8 * shards.size).order(ByteOrder.nativeOrder)
    shards.foreach(s => hs.putLong(s.handle))
    NativeReplay.allgatherGroup(hs, shards.size, slot)
  }

  /** State `row` of shard `ofShard` as device `onShard` holds it after `exchange` (waits for that slot's exchange);
   *  64 bytes, little-endian, all-zero = None. */
  def peerState(shards: IndexedSeq[RecoveredSnapshot], onShard: Int, ofShard: Int, row: Long, slot: Int): ByteBuffer = {
    val st = ByteBuffer.allocateDirect(64).order(ByteOrder.LITTLE_ENDIAN)
    NativeReplay.gatheredRead(shards(onShard).handle, slot, ofShard, row, 1L, st)
    st
  }
}
