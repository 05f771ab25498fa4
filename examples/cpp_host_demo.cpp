// cpp_host_demo.cpp — the reference's Counter model written against the C++ host mirror (include/surge_replay.hpp):
// the model keeps its literal handleEvent (scaladsl TestBoundedContext.scala:77-89) and adds the replay declaration; the
// store serves getAggregateBytes (AggregateStateStoreKafkaStreams.scala:83-85) from the GPU fold.
//
//   g++ -std=c++17 -Iinclude examples/cpp_host_demo.cpp -Lsurge_amd -lsurge_replay -Wl,-rpath,$PWD/surge_amd -o /tmp/cpp_host_demo
//
// Every check compares the GPU-recovered bytes with writeState(events.foldLeft(None)(handleEvent)) computed by the
// literal handleEvent below, on the reference's own cases (PersistentActorSpec.scala:134-168, 431-464, 466-493;
// KafkaPartitionerSpec.scala:10-21).  Exit code 0 = all good, 2 = no GPU (the library has no CPU fallback).
#include <cstdio>
#include <variant>

#include "surge_replay.hpp"

namespace {

struct State {
  std::string aggregateId;
  int count;
  int version;
};

struct CountIncremented { std::string aggregateId; int incrementBy; int sequenceNumber; };
struct CountDecremented { std::string aggregateId; int decrementBy; int sequenceNumber; };
struct NoOpEvent { std::string aggregateId; int sequenceNumber; };
struct ExceptionThrowingEvent { std::string aggregateId; int sequenceNumber; };
using BaseTestEvent = std::variant<CountIncremented, CountDecremented, NoOpEvent, ExceptionThrowingEvent>;

struct CounterModel : surge::ReplayableCommandModel<State, BaseTestEvent> {
  // TestBoundedContext.scala:77-89, literally
  std::optional<State> handleEvent(const std::optional<State>& agg, const BaseTestEvent& evt) const override {
    const std::string& id = std::visit([](const auto& e) -> const std::string& { return e.aggregateId; }, evt);
    const State current = agg.value_or(State{id, 0, 0});
    if (const auto* e = std::get_if<CountIncremented>(&evt)) return State{current.aggregateId, current.count + e->incrementBy, e->sequenceNumber};
    if (const auto* e = std::get_if<CountDecremented>(&evt)) return State{current.aggregateId, current.count - e->decrementBy, e->sequenceNumber};
    if (std::holds_alternative<NoOpEvent>(evt)) return current;  // Some(current): a no-op still materialises the aggregate
    throw std::runtime_error("This is expected");
  }
  // the same handler as data: what each event type does to the fixed-width state
  surge_replay_schema eventAlgebra() const override {
    surge_replay_schema sc;
    surge_replay_default_schema(&sc);
    return sc;
  }
  surge_event16 encodeEvent(const BaseTestEvent& evt) const override {
    surge_event16 e;
    std::memset(&e, 0, sizeof(e));
    if (const auto* i = std::get_if<CountIncremented>(&evt)) { e.type = SURGE_EVT_INC; e.seq = i->sequenceNumber; e.p.i.arg = i->incrementBy; }
    else if (const auto* d = std::get_if<CountDecremented>(&evt)) { e.type = SURGE_EVT_DEC; e.seq = d->sequenceNumber; e.p.i.arg = d->decrementBy; }
    else if (const auto* n = std::get_if<NoOpEvent>(&evt)) { e.type = SURGE_EVT_NOOP; e.seq = n->sequenceNumber; }
    else { e.type = SURGE_EVT_THROW; e.seq = std::get<ExceptionThrowingEvent>(evt).sequenceNumber; }
    return e;
  }
  std::string aggregateIdOf(const BaseTestEvent& evt) const override {
    return std::visit([](const auto& e) { return e.aggregateId; }, evt);
  }
  State stateFromFixed(const std::string& id, const surge_state64& s) const override { return State{id, s.count, s.version}; }
};

// Json.toJson(state) with play-json's compact printer (scaladsl TestBoundedContext.scala:125-129); ids here are plain ASCII
struct CounterFormat : surge::SurgeAggregateWriteFormatting<State> {
  surge::SerializedAggregate writeState(const State& s) const override {
    const std::string js = "{\"aggregateId\":\"" + s.aggregateId + "\",\"count\":" + std::to_string(s.count) + ",\"version\":" + std::to_string(s.version) + "}";
    return surge::SerializedAggregate{std::vector<uint8_t>(js.begin(), js.end()), {}};
  }
};

int fails = 0;
void check(bool ok, const char* what) {
  std::printf("%s  %s\n", ok ? "PASS" : "FAIL", what);
  if (!ok) ++fails;
}

}  // namespace

int main() {
  // KafkaPartitionerSpec: keys that share the text before ':' share a partition; known hash answers
  const surge::PartitionStringUpToColon partitioner;
  check(partitioner.partitionBy("agg-7:12") == "agg-7" && partitioner.partitionBy("noColon") == "noColon", "partitionBy = takeWhile(_ != ':')");
  check(partitioner.partitionForKey(partitioner.partitionBy("agg-7:12"), 64) == partitioner.partitionForKey("agg-7", 64), "same aggregate, same partition");
  // scala MurmurHash3.stringHash("") = 377927480, ("a") = -1454233464 (tests/test_oracle_kat.py pins them); abs(h % n)
  check(partitioner.partitionForKey("", 1000003) == 926349 && partitioner.partitionForKey("a", 1000003) == 229102, "abs(stringHash(key) % n) known answers");
  check(partitioner.partitionForKey(u8"id-\U0001F600", 7) == 3, "a non-BMP key hashes over its UTF-16 surrogate pair");

  auto model = std::make_shared<CounterModel>();
  auto fmt = std::make_shared<CounterFormat>();
  std::shared_ptr<surge::AggregateStateStore<State, BaseTestEvent>> store;
  try {
    store = std::make_shared<surge::AggregateStateStore<State, BaseTestEvent>>(model, fmt, 0);
  } catch (const surge::ReplayException& e) {
    if (e.status() == SURGE_E_DEVICE) { std::printf("no GPU: %s\n", e.what()); return 2; }
    std::printf("create failed: %s\n", e.what());
    return 1;
  }

  // an interleaved events topic: offset order across aggregates, per-aggregate order preserved
  std::vector<BaseTestEvent> topic;
  std::map<std::string, std::vector<BaseTestEvent>> perAggregate;
  auto publish = [&](BaseTestEvent e) {
    perAggregate[model->aggregateIdOf(e)].push_back(e);
    topic.push_back(std::move(e));
  };
  for (int i = 1; i <= 40; ++i) {
    publish(CountIncremented{"a", 1, i});
    if (i % 3 == 0) publish(CountDecremented{"b", 2, i});
    if (i % 5 == 0) publish(NoOpEvent{"c", i});
    if (i % 7 == 0) publish(CountIncremented{"b", i, i});
  }
  publish(CountIncremented{"poisoned", 1, 1});
  publish(ExceptionThrowingEvent{"poisoned", 2});
  publish(CountIncremented{"poisoned", 1, 3});
  store->restore(topic, /*capacity=*/16);

  bool all = true;
  for (const char* id : {"a", "b", "c"}) {
    const std::optional<State> want = model->applyEvents(std::nullopt, perAggregate[id]);
    const std::optional<std::vector<uint8_t>> got = store->getAggregateBytes(id);
    all = all && (want.has_value() == got.has_value()) && (!want || fmt->writeState(*want).value == *got);
  }
  check(all, "getAggregateBytes == writeState(events.foldLeft(None)(handleEvent)) for every aggregate");
  const std::optional<std::vector<uint8_t>> c = store->getAggregateBytes("c");
  check(c && std::string(c->begin(), c->end()) == "{\"aggregateId\":\"c\",\"count\":0,\"version\":0}", "only no-op events: materialised as State(id, 0, 0)");
  check(!store->getAggregateBytes("never-seen").has_value(), "unknown aggregate id is a KTable miss");
  bool threw = false;
  try { store->getAggregateBytes("poisoned"); } catch (const surge::AggregateInitializationException&) { threw = true; }
  check(threw, "an aggregate whose replay throws surfaces as an initialization failure");

  // streaming micro-batch on top of the recovered state (PersistentActorSpec.scala:134-168: (3,3) + 2 increments)
  std::vector<BaseTestEvent> batch = {CountIncremented{"a", 1, 41}, CountIncremented{"late", 5, 1}, CountIncremented{"a", 1, 42}};
  for (const auto& e : batch) perAggregate[model->aggregateIdOf(e)].push_back(e);
  store->applyEvents(batch);
  const std::optional<State> wantA = model->applyEvents(std::nullopt, perAggregate["a"]);
  check(store->getAggregateBytes("a") == fmt->writeState(*wantA).value && wantA->count == 42 && wantA->version == 42, "micro-batch folds onto the resident state");
  const std::optional<std::vector<uint8_t>> late = store->getAggregateBytes("late");
  check(late && std::string(late->begin(), late->end()) == "{\"aggregateId\":\"late\",\"count\":5,\"version\":1}", "an aggregate first seen in a micro-batch");

  // S1: the plugin's key/value store — recovered reads, later state-topic records overlay, tombstones delete
  surge::SurgeKafkaStreamsPersistencePlugin<State, BaseTestEvent> plugin{store};
  auto kv = plugin.createSupplier("aggregate-state");
  bool okKv = !plugin.enableLogging() && kv.get("a") == store->getAggregateBytes("a");
  kv.put("a", std::vector<uint8_t>{'x'});
  okKv = okKv && kv.get("a") == std::vector<uint8_t>{'x'};
  kv.put("a", std::nullopt);
  okKv = okKv && !kv.get("a").has_value();
  check(okKv, "createSupplier store: recovered read, overlay, tombstone");

  std::printf("%s\n", fails ? "FAILED" : "ALL PASS");
  return fails ? 1 : 0;
}
