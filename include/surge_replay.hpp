// surge_replay.hpp — header-only C++17 host layer above the C ABI (surge_replay.h), mirroring the reference's
// plugin interfaces for the replay path.  The reference is Scala (compiled JVM code) and no JVM exists in the
// build image, so this is the compiled-language mirror of the same contracts; the Python mirror in surge_amd/
// has the same names and behaviour and carries the test-suite.
//
//   surge::KafkaPartitionProvider / PartitionStringUpToColon
//       modules/common/src/main/scala/surge/kafka/KafkaPartitioner.scala:7-9, 38-42
//   surge::SerializedAggregate, SurgeAggregateWriteFormatting / ReadFormatting
//       modules/serialization/src/main/scala/surge/core/SerializedAggregate.scala:7, SurgeFormatting.scala:5-15
//   surge::AggregateCommandModel::handleEvent  (the semantic contract; the fold is CommandModels.scala:20,26)
//       modules/command-engine/scaladsl/src/main/scala/surge/scaladsl/command/CommandModels.scala:12-31
//   surge::AggregateStateStore::getAggregateBytes  (seam S2)
//       modules/common/src/main/scala/surge/kafka/streams/AggregateStateStoreKafkaStreams.scala:83-85
//   surge::SurgeKafkaStreamsPersistencePlugin::createSupplier  (seam S1)
//       modules/common/src/main/scala/surge/kafka/streams/SurgeKafkaStreamsPersistencePlugin.scala:12-15
#ifndef SURGE_REPLAY_HPP
#define SURGE_REPLAY_HPP

#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "surge_replay.h"

namespace surge {

// A failed Future on the JVM side: KTableInitializationSupport.fetchState retries it
// (modules/command-engine/core/src/main/scala/surge/internal/persistence/KTableInitializationSupport.scala:63-81).
class ReplayException : public std::runtime_error {
 public:
  ReplayException(int32_t status, const std::string& what) : std::runtime_error(what), status_(status) {}
  int32_t status() const { return status_; }

 private:
  int32_t status_;
};

// An aggregate whose replay hit an event whose handler throws (PersistentActor.scala:328-333).
class AggregateInitializationException : public std::runtime_error {
 public:
  using std::runtime_error::runtime_error;
};

struct SerializedAggregate {
  std::vector<uint8_t> value;
  std::map<std::string, std::string> headers;
};

struct SerializedMessage {
  std::string key;
  std::vector<uint8_t> value;
  std::map<std::string, std::string> headers;
};

template <class State>
struct SurgeAggregateReadFormatting {
  virtual ~SurgeAggregateReadFormatting() = default;
  virtual std::optional<State> readState(const std::vector<uint8_t>& bytes) const = 0;
};

template <class State>
struct SurgeAggregateWriteFormatting {
  virtual ~SurgeAggregateWriteFormatting() = default;
  virtual SerializedAggregate writeState(const State& state) const = 0;
};

// JVM strings are UTF-16: stringHash consumes code units, so UTF-8 keys are transcoded first.
inline std::vector<uint16_t> utf16_of(const std::string& s) {
  std::vector<uint16_t> out;
  size_t i = 0;
  while (i < s.size()) {
    uint32_t c = (uint8_t)s[i];
    int extra = c < 0x80 ? 0 : (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : -1;
    if (extra < 0 || i + (size_t)extra + 1 > s.size()) { out.push_back(0xFFFD); ++i; continue; }  // malformed: U+FFFD
    if (extra) c &= (0x3Fu >> extra);
    for (int k = 1; k <= extra; ++k) c = (c << 6) | ((uint8_t)s[i + (size_t)k] & 0x3F);
    i += (size_t)extra + 1;
    if (c >= 0x10000) {
      c -= 0x10000;
      out.push_back((uint16_t)(0xD800 + (c >> 10)));
      out.push_back((uint16_t)(0xDC00 + (c & 0x3FF)));
    } else {
      out.push_back((uint16_t)c);
    }
  }
  return out;
}

// trait KafkaPartitionProvider { def partitionForKey(partitionByString: String, numberOfPartitions: Int): Int }
struct KafkaPartitionProvider {
  virtual ~KafkaPartitionProvider() = default;
  // abs(MurmurHash3.stringHash(partitionByString) % numberOfPartitions): the whole string, as KafkaPartitioner.scala:8
  virtual int partitionForKey(const std::string& partitionByString, int numberOfPartitions) const {
    const std::vector<uint16_t> u = utf16_of(partitionByString);
    const int64_t off[2] = {0, (int64_t)u.size()};
    int32_t part = 0;
    const uint16_t dummy = 0;
    const int32_t rc = surge_replay_partition_hash(u.empty() ? &dummy : u.data(), off, 1, numberOfPartitions, &part);
    if (rc != SURGE_OK) throw ReplayException(rc, surge_replay_last_error(nullptr));
    return part;
  }
};

// final class PartitionStringUpToColon extends KafkaPartitioner[String] { partitionBy = _.takeWhile(_ != ':') }
struct PartitionStringUpToColon : KafkaPartitionProvider {
  std::string partitionBy(const std::string& key) const { return key.substr(0, key.find(':')); }
};

// trait AggregateCommandModel[Agg, Cmd, Evt] — the part of it on the replay path, plus the additive replay
// declaration (event algebra + fixed-width codecs) a model needs for GPU replay.
template <class Agg, class Evt>
struct ReplayableCommandModel {
  virtual ~ReplayableCommandModel() = default;
  virtual std::optional<Agg> handleEvent(const std::optional<Agg>& aggregate, const Evt& event) const = 0;
  virtual surge_replay_schema eventAlgebra() const = 0;
  virtual surge_event16 encodeEvent(const Evt& event) const = 0;
  virtual std::string aggregateIdOf(const Evt& event) const = 0;
  virtual Agg stateFromFixed(const std::string& aggregateId, const surge_state64& fixed) const = 0;
  // events.foldLeft(state)(handleEvent) — CommandModels.scala:26
  std::optional<Agg> applyEvents(std::optional<Agg> state, const std::vector<Evt>& events) const {
    for (const Evt& e : events) state = handleEvent(state, e);
    return state;
  }
};

// The GPU-backed aggregate state store behind getAggregateBytes (seam S2).
template <class Agg, class Evt>
class AggregateStateStore {
 public:
  AggregateStateStore(std::shared_ptr<const ReplayableCommandModel<Agg, Evt>> model,
                      std::shared_ptr<const SurgeAggregateWriteFormatting<Agg>> writeFormatting, int device = 0)
      : model_(std::move(model)), fmt_(std::move(writeFormatting)) {
    const surge_replay_schema sc = model_->eventAlgebra();
    const int32_t rc = surge_replay_create(&sc, device, &h_);
    if (rc != SURGE_OK) throw ReplayException(rc, surge_replay_last_error(nullptr));
  }
  ~AggregateStateStore() { surge_replay_destroy(h_); }
  AggregateStateStore(const AggregateStateStore&) = delete;
  AggregateStateStore& operator=(const AggregateStateStore&) = delete;

  // Recover from the events topic: records in offset order; `capacity` reserves room for aggregates that
  // only show up in later micro-batches.
  void restore(const std::vector<Evt>& eventsInOffsetOrder, size_t capacity = 0) {
    std::vector<int64_t> agg(eventsInOffsetOrder.size());
    for (size_t i = 0; i < eventsInOffsetOrder.size(); ++i) agg[i] = intern(model_->aggregateIdOf(eventsInOffsetOrder[i]));
    const size_t n_agg = std::max(keys_.size(), capacity);
    // stable group-by aggregate: counting sort keeps offset order inside an aggregate
    std::vector<int64_t> seg_off(n_agg + 1, 0);
    for (int64_t a : agg) ++seg_off[(size_t)a + 1];
    for (size_t a = 0; a < n_agg; ++a) seg_off[a + 1] += seg_off[a];
    std::vector<int64_t> cursor(seg_off.begin(), seg_off.end() - 1);
    std::vector<surge_event16> packed(eventsInOffsetOrder.size());
    for (size_t i = 0; i < eventsInOffsetOrder.size(); ++i) packed[(size_t)cursor[(size_t)agg[i]]++] = model_->encodeEvent(eventsInOffsetOrder[i]);
    check(surge_replay_load_csr(h_, seg_off.data(), (int64_t)n_agg, packed.data(), (int64_t)packed.size(), nullptr));
    check(surge_replay_fold(h_, SURGE_ALGO_AUTO));
    check(surge_replay_snapshot(h_, nullptr, nullptr));  // publishes the mirror that serves point reads
    n_agg_ = n_agg;
  }

  // Streaming micro-batch onto the resident state (PersistentActor.doApplyEvent for many aggregates at once).
  void applyEvents(const std::vector<Evt>& eventsInOffsetOrder) {
    std::vector<int64_t> agg(eventsInOffsetOrder.size());
    std::vector<surge_event16> enc(eventsInOffsetOrder.size());
    for (size_t i = 0; i < eventsInOffsetOrder.size(); ++i) {
      agg[i] = intern(model_->aggregateIdOf(eventsInOffsetOrder[i]));
      enc[i] = model_->encodeEvent(eventsInOffsetOrder[i]);
    }
    if (keys_.size() > n_agg_) {  // aggregates born after recovery: the resident state grows, new rows are None
      const size_t grown = std::max(keys_.size(), n_agg_ + n_agg_ / 2);
      check(surge_replay_grow(h_, (int64_t)grown));
      n_agg_ = grown;
    }
    check(surge_replay_append_events(h_, agg.data(), enc.data(), (int64_t)enc.size()));
    check(surge_replay_snapshot(h_, nullptr, nullptr));
  }

  // def getAggregateBytes(aggregateId: String): Future[Option[Array[Byte]]]
  // nullopt = no such aggregate (KTable miss) or tombstone; else writeState(state).value
  std::optional<std::vector<uint8_t>> getAggregateBytes(const std::string& aggregateId) const {
    const std::optional<Agg> a = getAggregate(aggregateId);
    if (!a) return std::nullopt;
    return fmt_->writeState(*a).value;
  }

  std::optional<Agg> getAggregate(const std::string& aggregateId) const {
    const auto it = index_.find(aggregateId);
    if (it == index_.end() || (size_t)it->second >= n_agg_) return std::nullopt;
    surge_state64 st;
    uint8_t present = 0;
    check(surge_replay_get(h_, it->second, &st, &present));
    if (st.flags & SURGE_STATE_POISONED)
      throw AggregateInitializationException("replay of aggregate " + aggregateId + " hit an event whose handler throws");
    if (!present) return std::nullopt;
    return model_->stateFromFixed(aggregateId, st);
  }

  surge_replay_handle* handle() const { return h_; }

 private:
  void check(int32_t rc) const {
    if (rc != SURGE_OK) throw ReplayException(rc, surge_replay_last_error(h_));
  }
  int64_t intern(const std::string& id) {
    const auto it = index_.find(id);
    if (it != index_.end()) return it->second;
    const int64_t i = (int64_t)keys_.size();
    keys_.push_back(id);
    index_.emplace(id, i);
    return i;
  }

  std::shared_ptr<const ReplayableCommandModel<Agg, Evt>> model_;
  std::shared_ptr<const SurgeAggregateWriteFormatting<Agg>> fmt_;
  surge_replay_handle* h_ = nullptr;
  std::vector<std::string> keys_;
  std::unordered_map<std::string, int64_t> index_;
  size_t n_agg_ = 0;
};

// The key/value store the KTable topology writes state-topic records into (seam S1): reads fall through to the
// GPU-recovered store, puts (later state-topic records) overlay it — last write wins, an empty optional is a
// tombstone (SurgeStateStoreConsumer.scala:69).
template <class Agg, class Evt>
class GpuReplayKeyValueStore {
 public:
  GpuReplayKeyValueStore(std::string name, std::shared_ptr<const AggregateStateStore<Agg, Evt>> recovered)
      : name_(std::move(name)), recovered_(std::move(recovered)) {}
  void put(const std::string& key, std::optional<std::vector<uint8_t>> value) { overlay_[key] = std::move(value); }
  std::optional<std::vector<uint8_t>> get(const std::string& key) const {
    const auto it = overlay_.find(key);
    if (it != overlay_.end()) return it->second;
    return recovered_ ? recovered_->getAggregateBytes(key) : std::nullopt;
  }
  const std::string& name() const { return name_; }

 private:
  std::string name_;
  std::shared_ptr<const AggregateStateStore<Agg, Evt>> recovered_;
  std::map<std::string, std::optional<std::vector<uint8_t>>> overlay_;
};

// trait SurgeKafkaStreamsPersistencePlugin { def createSupplier(storeName: String); def enableLogging: Boolean }
template <class Agg, class Evt>
struct SurgeKafkaStreamsPersistencePlugin {
  std::shared_ptr<const AggregateStateStore<Agg, Evt>> recovered;
  // the store is rebuilt from the events topic, so it needs no changelog
  bool enableLogging() const { return false; }
  GpuReplayKeyValueStore<Agg, Evt> createSupplier(const std::string& storeName) const {
    return GpuReplayKeyValueStore<Agg, Evt>(storeName, recovered);
  }
};

}  // namespace surge

#endif  // SURGE_REPLAY_HPP
