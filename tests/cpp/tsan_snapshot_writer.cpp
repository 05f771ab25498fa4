// TEST INFRASTRUCTURE: the snapshot writer frames partitions on several host threads; run it under ThreadSanitizer.
//   g++ -std=c++17 -O1 -g -fsanitize=thread -Iinclude tests/cpp/tsan_snapshot_writer.cpp surge_amd/csrc/snapshot_writer.cpp \
//       surge_amd/csrc/lz4_frame.cpp surge_amd/csrc/ingest.cpp -lpthread -o tsan_snapshot_writer && ./tsan_snapshot_writer
// Also checks the result: every partition's batches decode (through the product's own reader) to the records appended,
// in index order, for both codecs.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "surge_ingest.h"
#include "surge_replay.h"
#include "surge_snapshot.h"

int main() {
  const int64_t n = 200000;
  const int32_t P = 64;
  std::vector<uint8_t> kind((size_t)n), keys, vals;
  std::vector<int32_t> part((size_t)n);
  std::vector<int64_t> key_off(1, 0), val_off(1, 0);
  for (int64_t i = 0; i < n; ++i) {
    kind[(size_t)i] = (i % 11 == 0) ? SURGE_SNAP_SKIP : (i % 7 == 0 ? SURGE_SNAP_TOMBSTONE : SURGE_SNAP_VALUE);
    part[(size_t)i] = (int32_t)((i * 2654435761u) % P);
    char k[32], v[96];
    const int kl = std::snprintf(k, sizeof(k), "acct-%08lld", (long long)i);
    const int vl = std::snprintf(v, sizeof(v), "{\"aggregateId\":\"acct-%08lld\",\"count\":%lld,\"version\":%lld}", (long long)i, (long long)(i % 97), (long long)i);
    keys.insert(keys.end(), k, k + kl);
    vals.insert(vals.end(), v, v + vl);
    key_off.push_back((int64_t)keys.size());
    val_off.push_back((int64_t)vals.size());
  }
  int fails = 0;
  for (const int32_t codec : {SURGE_SNAPSHOT_CODEC_NONE, SURGE_SNAPSHOT_CODEC_LZ4}) {
    surge_snapshot_writer* w = nullptr;
    if (surge_snapshot_writer_create(P, 1000, 0, &w) != 0 || surge_snapshot_writer_set_compression(w, codec) != 0) return 1;
    for (int rep = 0; rep < 2; ++rep)  // two appends: the logs continue
      if (surge_snapshot_writer_append(w, n, kind.data(), part.data(), keys.data(), key_off.data(), vals.data(), val_off.data(), 1700000000000ll + rep) != 0) return 1;
    if (surge_snapshot_writer_flush(w) != 0) return 1;
    int64_t total = 0;
    for (int32_t p = 0; p < P; ++p) {
      const uint8_t* data = nullptr;
      int64_t len = 0, n_rec = 0, next = 0;
      if (surge_snapshot_writer_partition(w, p, &data, &len, &n_rec, &next) != 0) return 1;
      surge_ingest* g = nullptr;
      int64_t consumed = 0;
      if (surge_ingest_create(SURGE_INGEST_READ_COMMITTED, &g) != 0) return 1;
      if (surge_ingest_feed(g, data, len, &consumed) != 0 || consumed != len || surge_ingest_ready(g) != n_rec) ++fails;
      std::vector<surge_ingest_record> recs((size_t)n_rec);
      int64_t got = 0;
      if (surge_ingest_drain(g, n_rec, recs.data(), &got) != 0 || got != n_rec) ++fails;
      const uint8_t* arena = surge_ingest_arena(g);
      int64_t r = 0;
      for (int rep = 0; rep < 2 && !fails; ++rep)
        for (int64_t i = 0; i < n; ++i) {
          if (kind[(size_t)i] == SURGE_SNAP_SKIP || part[(size_t)i] != p) continue;
          const surge_ingest_record& x = recs[(size_t)r];
          const int64_t kl = key_off[(size_t)i + 1] - key_off[(size_t)i], vl = val_off[(size_t)i + 1] - val_off[(size_t)i];
          bool ok = x.offset == r && x.key_len == kl && std::memcmp(arena + x.key_off, keys.data() + key_off[(size_t)i], (size_t)kl) == 0;
          if (kind[(size_t)i] == SURGE_SNAP_TOMBSTONE) ok = ok && x.value_len == -1;
          else ok = ok && x.value_len == vl && std::memcmp(arena + x.value_off, vals.data() + val_off[(size_t)i], (size_t)vl) == 0;
          if (!ok) { ++fails; break; }
          ++r;
        }
      if (r != n_rec) ++fails;
      total += n_rec;
      surge_ingest_destroy(g);
    }
    std::printf("codec %d: %lld records in %d partitions, %s\n", codec, (long long)total, P, fails ? "FAIL" : "PASS");
    surge_snapshot_writer_destroy(w);
  }
  return fails ? 1 : 0;
}
