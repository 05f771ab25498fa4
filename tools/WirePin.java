// tools/WirePin.java — generates the kafka-clients pin for the events-topic wire format (SURVEY §8c: "parity unpinned").
//
//   javac -cp kafka-clients-2.8.x.jar:lz4-java-1.7.x.jar tools/WirePin.java -d /tmp/wirepin
//   java  -cp /tmp/wirepin:kafka-clients-2.8.x.jar:lz4-java-1.7.x.jar:slf4j-api.jar WirePin > tests/golden/wire_pin.tsv
//
// (kafka-clients as the reference's build pins it: project/Dependencies.scala; the producer settings it mirrors:
// modules/common/src/main/resources/reference.conf:111-126 — compression.type lz4, message format v2.)
// One line per case:  <name> TAB <hex of the record batches kafka-clients wrote> TAB <records as key-hex:value-hex, comma separated, "-" = null>
// tests/test_ingest.py::test_kafka_clients_pin_of_the_wire_format feeds the bytes to surge_ingest and compares the records it
// delivers with the third column; the test is skipped until a JDK with kafka-clients has produced the file (none in the build image).
import java.nio.ByteBuffer;
import java.nio.charset.StandardCharsets;
import java.util.ArrayList;
import java.util.List;

import org.apache.kafka.common.record.CompressionType;
import org.apache.kafka.common.record.ControlRecordType;
import org.apache.kafka.common.record.EndTransactionMarker;
import org.apache.kafka.common.record.MemoryRecords;
import org.apache.kafka.common.record.MemoryRecordsBuilder;
import org.apache.kafka.common.record.RecordBatch;
import org.apache.kafka.common.record.SimpleRecord;
import org.apache.kafka.common.record.TimestampType;

public class WirePin {
  static String hex(byte[] b) {
    if (b == null) return "-";
    StringBuilder sb = new StringBuilder();
    for (byte x : b) sb.append(String.format("%02x", x & 0xff));
    return sb.toString();
  }

  static String hex(ByteBuffer buf) {
    ByteBuffer d = buf.duplicate();
    byte[] b = new byte[d.remaining()];
    d.get(b);
    return hex(b);
  }

  static SimpleRecord event(int agg, int seq, int by) {
    String id = String.format("acct-%08d", agg);
    String key = id + ":" + seq;
    String value = "{\"aggregateId\":\"" + id + "\",\"incrementBy\":" + by + ",\"sequenceNumber\":" + seq + ",\"_type\":\"countIncremented\"}";
    return new SimpleRecord(1700000000000L + seq, key.getBytes(StandardCharsets.UTF_8), value.getBytes(StandardCharsets.UTF_8));
  }

  static void emit(String name, ByteBuffer wire, List<SimpleRecord> delivered) {
    StringBuilder recs = new StringBuilder();
    for (SimpleRecord r : delivered) {
      if (recs.length() > 0) recs.append(',');
      byte[] k = r.key() == null ? null : new byte[r.key().remaining()];
      byte[] v = r.value() == null ? null : new byte[r.value().remaining()];
      if (k != null) r.key().duplicate().get(k);
      if (v != null) r.value().duplicate().get(v);
      recs.append(hex(k)).append(':').append(hex(v));
    }
    System.out.println(name + "\t" + hex(wire) + "\t" + recs);
  }

  public static void main(String[] args) {
    // 1. a plain lz4 batch and the same records uncompressed
    for (CompressionType ct : new CompressionType[] {CompressionType.LZ4, CompressionType.NONE}) {
      List<SimpleRecord> recs = new ArrayList<>();
      for (int i = 0; i < 200; ++i) recs.add(event(i % 7, i / 7 + 1, i % 1000));
      MemoryRecords mr = MemoryRecords.withRecords(RecordBatch.MAGIC_VALUE_V2, 100L, ct, TimestampType.CREATE_TIME, recs.toArray(new SimpleRecord[0]));
      emit("plain_" + ct.name, mr.buffer(), recs);
    }
    // 2. a batch large enough for several 64 KiB LZ4 blocks
    {
      List<SimpleRecord> recs = new ArrayList<>();
      for (int i = 0; i < 3000; ++i) recs.add(event(i, 1, (i * 7919) % 1000));
      MemoryRecords mr = MemoryRecords.withRecords(RecordBatch.MAGIC_VALUE_V2, 0L, CompressionType.LZ4, TimestampType.CREATE_TIME, recs.toArray(new SimpleRecord[0]));
      emit("several_blocks_lz4", mr.buffer(), recs);
    }
    // 3. one committed and one aborted transaction, as a transactional producer's partition log holds them
    {
      ByteBuffer out = ByteBuffer.allocate(1 << 20);
      List<SimpleRecord> delivered = new ArrayList<>();
      long offset = 0;
      for (int txn = 0; txn < 2; ++txn) {
        List<SimpleRecord> recs = new ArrayList<>();
        for (int i = 0; i < 50; ++i) recs.add(event(i % 5, txn * 10 + i / 5 + 1, i));
        MemoryRecordsBuilder b = MemoryRecords.builder(ByteBuffer.allocate(1 << 16), RecordBatch.MAGIC_VALUE_V2, CompressionType.LZ4, TimestampType.CREATE_TIME, offset,
            1700000000000L, 4242L, (short) 0, txn * 50, true, RecordBatch.NO_PARTITION_LEADER_EPOCH);
        for (SimpleRecord r : recs) b.append(r);
        out.put(b.build().buffer());
        offset += recs.size();
        ControlRecordType kind = txn == 0 ? ControlRecordType.COMMIT : ControlRecordType.ABORT;
        out.put(MemoryRecords.withEndTransactionMarker(offset, 1700000000050L, RecordBatch.NO_PARTITION_LEADER_EPOCH, 4242L, (short) 0, new EndTransactionMarker(kind, 0)).buffer());
        offset += 1;
        if (txn == 0) delivered.addAll(recs);
      }
      out.flip();
      emit("transactions_commit_then_abort", out, delivered);
    }
    // 4. the flush record (empty key, empty value: KafkaProducerActorImpl.scala:322-329) and a null value
    {
      SimpleRecord[] recs = {new SimpleRecord(0L, new byte[0], new byte[0]), event(1, 1, 5)};
      MemoryRecords mr = MemoryRecords.withRecords(RecordBatch.MAGIC_VALUE_V2, 7L, CompressionType.LZ4, TimestampType.CREATE_TIME, recs);
      List<SimpleRecord> delivered = new ArrayList<>();
      delivered.add(recs[1]);
      emit("flush_record_lz4", mr.buffer(), delivered);
    }
  }
}
