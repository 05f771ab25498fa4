// TEST INFRASTRUCTURE: surge_ingest_group under ThreadSanitizer — the framing pool (8 threads, one job per feed), the six
// rotating slabs read by a consumer thread while the next feeds are framed (what a device push does with them), and the
// undo of a failed feed.
//   g++ -std=c++17 -O1 -g -fsanitize=thread -Iinclude tests/cpp/tsan_ingest_group.cpp surge_amd/csrc/ingest.cpp \
//       surge_amd/csrc/event_decode.cpp surge_amd/csrc/f64_text.cpp surge_amd/csrc/lz4_frame.cpp -lpthread -o tsan_ingest_group
// The topic comes from the independent test-side writer (tests/native/wire_writer.c: transactions per flush, aborted
// flushes, markers a fetch late).  Checked as well as raced: every fetch's sections == what each partition's own framer
// (a plain surge_ingest, fed alone) delivers for it.
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "surge_ingest.h"
#include "surge_replay.h"

extern "C" {
#include "../native/wire_writer.c"
}

namespace {

struct Fetch {
  std::vector<std::string> part;  // bytes per partition
};

struct Framed {
  std::vector<surge_batch_section> sec;
  const uint8_t* slab = nullptr;
};

uint64_t digest(const std::vector<surge_batch_section>& sec, const uint8_t* base) {
  uint64_t h = 1469598103934665603ull;
  for (const surge_batch_section& s : sec) {
    h = (h ^ (uint64_t)s.base_offset) * 1099511628211ull;
    h = (h ^ (uint64_t)s.n_records) * 1099511628211ull;
    h = (h ^ (uint64_t)s.codec) * 1099511628211ull;
    for (int64_t i = 0; i < s.byte_len; ++i) h = (h ^ base[s.byte_off + i]) * 1099511628211ull;
  }
  return h;
}

}  // namespace

int main() {
  const int32_t P = 16;
  const int F = 14;  // more than two turns of the six slabs
  // ---- the topic ----------------------------------------------------------------------------------------------------------
  std::vector<Fetch> fetches;
  {
    wire_topic* t = surge_test_wire_topic_create(P);
    int64_t counts[8] = {0};
    uint64_t rng = 88172645463325252ull;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    for (int f = 0; f <= F; ++f) {
      const int64_t n = f < F ? 6000 + 500 * f : 0;  // the last "fetch" only carries the markers held back
      std::vector<int32_t> part((size_t)n);
      std::vector<uint8_t> keys, vals;
      std::vector<int64_t> ko(1, 0), vo(1, 0);
      for (int64_t i = 0; i < n; ++i) {
        part[(size_t)i] = (int32_t)(next() % P);
        char k[40], v[128];
        const int kl = std::snprintf(k, sizeof k, "acct-%08d:%d", (int)(next() % 5000), (int)i);
        const int vl = std::snprintf(v, sizeof v, "{\"aggregateId\":\"acct-%08d\",\"incrementBy\":%d,\"sequenceNumber\":%d,\"_type\":\"countIncremented\"}",
                                     (int)(next() % 5000), (int)(next() % 1000), (int)i);
        keys.insert(keys.end(), k, k + kl);
        vals.insert(vals.end(), v, v + vl);
        ko.push_back((int64_t)keys.size());
        vo.push_back((int64_t)vals.size());
      }
      keys.push_back(0);
      vals.push_back(0);
      if (surge_test_wire_topic_fetch(t, n, part.data(), keys.data(), ko.data(), vals.data(), vo.data(), 48, 4096, f % 3 != 2, 5, 4, counts) != 0) return 2;
      Fetch fe;
      for (int32_t p = 0; p < P; ++p) {
        int64_t len = 0;
        const uint8_t* d = surge_test_wire_topic_partition(t, p, &len);
        fe.part.emplace_back((const char*)d, (size_t)len);
      }
      fetches.push_back(std::move(fe));
    }
    surge_test_wire_topic_destroy(t);
    if (counts[1] == 0 || counts[3] == 0) return 2;
  }
  // ---- the group: this thread frames, a consumer thread reads up to five fetches behind ------------------------------------------
  surge_ingest_group* grp = nullptr;
  if (surge_ingest_group_create(P, SURGE_INGEST_READ_COMMITTED | SURGE_INGEST_DEVICE_LZ4, &grp) != 0) return 2;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Framed> ready;
  size_t consumed_fetches = 0;
  bool done = false;
  int fails = 0;
  std::vector<uint64_t> got(fetches.size(), 0);
  std::thread consumer([&] {
    size_t f = 0;
    for (;;) {
      Framed fr;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done || !ready.empty(); });
        if (ready.empty()) return;
        fr = std::move(ready.front());
        ready.pop_front();
      }
      got[f] = digest(fr.sec, fr.slab);
      ++f;
      {
        std::lock_guard<std::mutex> lk(mu);
        consumed_fetches = f;
      }
      cv.notify_all();
    }
  });
  for (size_t f = 0; f < fetches.size(); ++f) {
    {  // a slab is written again six feeds later: at most five fetches may still be unread when the next one is framed
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return f < 5 + consumed_fetches; });
    }
    std::vector<const uint8_t*> data((size_t)P);
    std::vector<int64_t> len((size_t)P), consumed((size_t)P);
    for (int32_t p = 0; p < P; ++p) {
      data[(size_t)p] = (const uint8_t*)fetches[f].part[(size_t)p].data();
      len[(size_t)p] = (int64_t)fetches[f].part[(size_t)p].size();
    }
    Framed fr;
    fr.sec.resize(8192);
    int64_t n_sec = 0;
    if (f % 4 == 1) {  // first a response with a flipped byte in one partition, then a table that is too small: both are undone
      std::string bad = fetches[f].part[3];
      if (!bad.empty()) {
        bad[bad.size() / 2] = (char)(bad[bad.size() / 2] ^ 0x5a);
        std::vector<const uint8_t*> d2 = data;
        d2[3] = (const uint8_t*)bad.data();
        const int32_t rc = surge_ingest_group_feed(grp, d2.data(), len.data(), 8, consumed.data(), 8192, fr.sec.data(), &n_sec, &fr.slab);
        if (rc == 0 || n_sec != 0) ++fails;
      }
      const int32_t rc2 = surge_ingest_group_feed(grp, data.data(), len.data(), 8, consumed.data(), 1, fr.sec.data(), &n_sec, &fr.slab);
      if (rc2 != SURGE_E_INVALID || n_sec < 2) ++fails;
    }
    const int32_t rc = surge_ingest_group_feed(grp, data.data(), len.data(), 8, consumed.data(), 8192, fr.sec.data(), &n_sec, &fr.slab);
    if (rc != 0) {
      std::printf("feed %zu: %d %s\n", f, rc, surge_ingest_group_last_error(grp));
      ++fails;
      break;
    }
    for (int32_t p = 0; p < P; ++p)
      if (consumed[(size_t)p] != len[(size_t)p]) ++fails;
    fr.sec.resize((size_t)n_sec);
    {
      std::lock_guard<std::mutex> lk(mu);
      ready.push_back(std::move(fr));
    }
    cv.notify_all();
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    done = true;
  }
  cv.notify_all();
  consumer.join();
  // ---- what every partition's own framer (a plain surge_ingest, fed alone) delivers, fetch by fetch ----------------------------
  std::vector<uint64_t> want_flat(fetches.size(), 0);
  {
    std::vector<surge_ingest*> single((size_t)P, nullptr);
    for (int32_t p = 0; p < P; ++p)
      if (surge_ingest_create(SURGE_INGEST_READ_COMMITTED | SURGE_INGEST_FRAMES | SURGE_INGEST_DEVICE_LZ4, &single[(size_t)p]) != 0) return 2;
    for (size_t f = 0; f < fetches.size(); ++f) {
      // (the digest is a running FNV over the sections, partition after partition: the order a group delivers them in)
      uint64_t h = 1469598103934665603ull;
      for (int32_t p = 0; p < P; ++p) {
        const std::string& b = fetches[f].part[(size_t)p];
        int64_t consumed = 0;
        surge_ingest_feed(single[(size_t)p], (const uint8_t*)b.data(), (int64_t)b.size(), &consumed);
        std::vector<surge_batch_section> sec(4096);
        int64_t n = 0;
        surge_ingest_drain_sections(single[(size_t)p], 4096, sec.data(), &n);
        const uint8_t* base = surge_ingest_arena(single[(size_t)p]);
        for (int64_t k = 0; k < n; ++k) {
          const surge_batch_section& s = sec[(size_t)k];
          h = (h ^ (uint64_t)s.base_offset) * 1099511628211ull;
          h = (h ^ (uint64_t)s.n_records) * 1099511628211ull;
          h = (h ^ (uint64_t)s.codec) * 1099511628211ull;
          for (int64_t i = 0; i < s.byte_len; ++i) h = (h ^ base[s.byte_off + i]) * 1099511628211ull;
        }
      }
      want_flat[f] = h;
    }
    for (surge_ingest* g : single) surge_ingest_destroy(g);
  }
  int64_t c[8];
  surge_ingest_group_counters(grp, c);
  surge_ingest_group_destroy(grp);
  for (size_t f = 0; f < fetches.size(); ++f)
    if (got[f] != want_flat[f]) {
      std::printf("fetch %zu: sections differ from the partitions' own framers\n", f);
      ++fails;
    }
  if (c[7] != 0 || c[3] == 0 || c[4] == 0) ++fails;  // no transaction left open; aborted records and markers were seen
  std::printf(fails ? "FAIL %d\n" : "PASS group: %lld batches, %lld control, %lld records delivered, %lld aborted\n", fails ? fails : 0, (long long)c[0], (long long)c[4],
              (long long)c[2], (long long)c[3]);
  return fails ? 1 : 0;
}
