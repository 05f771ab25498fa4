// cpp_persistence_demo.cpp — the callers of the fold through the C++ host mirror
// (include/surge_persistence.hpp), held to the literal expectations of the reference's own specs with the state of record
// on the GPU:
//   PersistentActorSpec.scala :134-168 (increment on (3,3): one publish of event + state, (4,4)), :218-226 (initialise from
//   the store), :229-288 (nothing published when nothing changed), :292-330 (publishStateOnly), :431-464 (domain exceptions
//   -> ACKError, actor still usable), :466-493 (commands one at a time), :495-508 (no-op events are published), :512-529
//   (ApplyEvents publishes only state records); KafkaProducerActorImplSpec.scala :343-368, :692-705 (isAggregateStateCurrent
//   follows KTable progress; newest in-flight offset per key); KTableInitializationSupport.scala:37-81 (retry intervals, 10
//   attempts); MultilanguageGatewayServiceImplSpec.scala:72-135 (new aggregate: (1,1), (2,2), decrement -> (1,3)).
//
//   g++ -std=c++17 -Iinclude examples/cpp_persistence_demo.cpp -Lsurge_amd -lsurge_replay -Wl,-rpath,$PWD/surge_amd -o /tmp/cpp_persistence_demo
//
// Exit code 0 = all good, 2 = no GPU (the CPU-only protocol checks still run and print first; no CPU fallback for the store).
#include <cstdio>

#include "surge_persistence.hpp"

namespace {

// ---- scaladsl TestBoundedContext.scala:15-134 -------------------------------------------------------------------------------
struct State {
  std::string aggregateId;
  int count;
  int version;
};
struct CountIncremented { std::string aggregateId; int incrementBy; int sequenceNumber; };
struct CountDecremented { std::string aggregateId; int decrementBy; int sequenceNumber; };
struct NoOpEvent { std::string aggregateId; int sequenceNumber; };
struct ExceptionThrowingEvent { std::string aggregateId; int sequenceNumber; std::string message; };
using Event = std::variant<CountIncremented, CountDecremented, NoOpEvent, ExceptionThrowingEvent>;

struct Increment { std::string aggregateId; };
struct Decrement { std::string aggregateId; };
struct DoNothing { std::string aggregateId; };
struct CreateNoOpEvent { std::string aggregateId; };
struct FailCommandProcessing { std::string aggregateId; std::string message; };
struct CreateExceptionThrowingEvent { std::string aggregateId; std::string message; };
using Command = std::variant<Increment, Decrement, DoNothing, CreateNoOpEvent, FailCommandProcessing, CreateExceptionThrowingEvent>;

struct CounterModel : surge::AggregateCommandModel<State, Command, Event> {
  // :91-106
  std::vector<Event> processCommand(const std::optional<State>& agg, const Command& cmd) const override {
    const int seq = (agg ? agg->version : 0) + 1;
    if (const auto* c = std::get_if<Increment>(&cmd)) return {CountIncremented{c->aggregateId, 1, seq}};
    if (const auto* c = std::get_if<Decrement>(&cmd)) return {CountDecremented{c->aggregateId, 1, seq}};
    if (const auto* c = std::get_if<CreateNoOpEvent>(&cmd)) return {NoOpEvent{c->aggregateId, seq}};
    if (std::holds_alternative<DoNothing>(cmd)) return {};
    if (const auto* c = std::get_if<FailCommandProcessing>(&cmd)) throw std::runtime_error(c->message);
    const auto& c = std::get<CreateExceptionThrowingEvent>(cmd);
    return {ExceptionThrowingEvent{c.aggregateId, seq, c.message}};
  }
  // :77-89
  std::optional<State> handleEvent(const std::optional<State>& agg, const Event& evt) const override {
    const std::string& id = std::visit([](const auto& e) -> const std::string& { return e.aggregateId; }, evt);
    const State current = agg.value_or(State{id, 0, 0});
    if (const auto* e = std::get_if<CountIncremented>(&evt)) return State{current.aggregateId, current.count + e->incrementBy, e->sequenceNumber};
    if (const auto* e = std::get_if<CountDecremented>(&evt)) return State{current.aggregateId, current.count - e->decrementBy, e->sequenceNumber};
    if (std::holds_alternative<NoOpEvent>(evt)) return current;
    throw std::runtime_error(std::get<ExceptionThrowingEvent>(evt).message);
  }
  surge_replay_schema eventAlgebra() const override {
    surge_replay_schema sc;
    surge_replay_default_schema(&sc);
    return sc;
  }
  surge_event16 encodeEvent(const Event& evt) const override {
    surge_event16 e;
    std::memset(&e, 0, sizeof(e));
    if (const auto* i = std::get_if<CountIncremented>(&evt)) { e.type = SURGE_EVT_INC; e.seq = i->sequenceNumber; e.p.i.arg = i->incrementBy; }
    else if (const auto* d = std::get_if<CountDecremented>(&evt)) { e.type = SURGE_EVT_DEC; e.seq = d->sequenceNumber; e.p.i.arg = d->decrementBy; }
    else if (const auto* n = std::get_if<NoOpEvent>(&evt)) { e.type = SURGE_EVT_NOOP; e.seq = n->sequenceNumber; }
    else { e.type = SURGE_EVT_THROW; e.seq = std::get<ExceptionThrowingEvent>(evt).sequenceNumber; }
    return e;
  }
  std::string aggregateIdOf(const Event& evt) const override {
    return std::visit([](const auto& e) { return e.aggregateId; }, evt);
  }
  State stateFromFixed(const std::string& id, const surge_state64& s) const override { return State{id, s.count, s.version}; }
};

std::string jsonOf(const State& s) {
  return "{\"aggregateId\":\"" + s.aggregateId + "\",\"count\":" + std::to_string(s.count) + ",\"version\":" + std::to_string(s.version) + "}";
}

// Json.toJson(state) / Json.parse(bytes).as[State] (:125-134), compact printer, plain-ASCII ids
struct CounterFormat : surge::SurgeAggregateWriteFormatting<State>, surge::SurgeAggregateReadFormatting<State> {
  surge::SerializedAggregate writeState(const State& s) const override {
    const std::string js = jsonOf(s);
    return surge::SerializedAggregate{std::vector<uint8_t>(js.begin(), js.end()), {}};
  }
  std::optional<State> readState(const std::vector<uint8_t>& bytes) const override {
    const std::string js(bytes.begin(), bytes.end());
    State s;
    char id[128];
    if (std::sscanf(js.c_str(), "{\"aggregateId\":\"%127[^\"]\",\"count\":%d,\"version\":%d}", id, &s.count, &s.version) != 3) return std::nullopt;
    s.aggregateId = id;
    return s;
  }
};

// key = "aggregateId:sequenceNumber" (:108-123); the value's text is the test fixture's own JSON
struct CounterEventFormat : surge::SurgeEventWriteFormatting<Event> {
  surge::SerializedMessage writeEvent(const Event& evt) const override {
    const std::string id = std::visit([](const auto& e) { return e.aggregateId; }, evt);
    const int seq = std::visit([](const auto& e) { return e.sequenceNumber; }, evt);
    std::string js = "{\"aggregateId\":\"" + id + "\",\"sequenceNumber\":" + std::to_string(seq) + "}";
    return surge::SerializedMessage{id + ":" + std::to_string(seq), std::vector<uint8_t>(js.begin(), js.end()), {}};
  }
};

using Store = surge::AggregateStateStore<State, Event>;
using Publisher = surge::StatePublisher<Store, Event>;
using Actor = surge::GpuPersistentActor<State, Command, Event, Store>;

int fails = 0;
void check(bool ok, const char* what) {
  std::printf("%s  %s\n", ok ? "PASS" : "FAIL", what);
  if (!ok) ++fails;
}

bool isState(const surge::Ack<State>& a, const std::string& id, int count, int version) {
  const auto* s = std::get_if<surge::ACKSuccess<State>>(&a);
  return s && s->aggregateState && s->aggregateState->aggregateId == id && s->aggregateState->count == count && s->aggregateState->version == version;
}
bool isError(const surge::Ack<State>& a, const std::string& what) {
  const auto* e = std::get_if<surge::ACKError>(&a);
  return e && e->what == what;
}
std::string text(const std::optional<std::vector<uint8_t>>& v) { return v ? std::string(v->begin(), v->end()) : std::string("<null>"); }

// a store for the retry logic alone (S2 only, no GPU)
struct FakeStore {
  std::map<std::string, std::vector<uint8_t>> values;
  int failures = 0, reads = 0;
  std::optional<std::vector<uint8_t>> getAggregateBytes(const std::string& id) {
    ++reads;
    if (failures > 0) { --failures; throw std::runtime_error("InvalidStateStoreException: rebalancing"); }
    const auto it = values.find(id);
    return it == values.end() ? std::nullopt : std::optional<std::vector<uint8_t>>(it->second);
  }
  void applyEvents(const std::vector<Event>&) {}
};

std::shared_ptr<surge::SurgeCommandBusinessLogic<State, Command, Event>> businessLogic(bool publishStateOnly = false) {
  auto fmt = std::make_shared<CounterFormat>();
  auto bl = std::make_shared<surge::SurgeCommandBusinessLogic<State, Command, Event>>();
  bl->aggregateName = "CountAggregate";
  bl->stateTopic = {"testStateTopic"};
  bl->eventsTopic = {"testEventsTopic"};
  bl->commandModel = std::make_shared<CounterModel>();
  bl->aggregateReadFormatting = fmt;
  bl->aggregateWriteFormatting = fmt;
  bl->eventWriteFormatting = std::make_shared<CounterEventFormat>();
  bl->publishStateOnly = publishStateOnly;
  return bl;
}

}  // namespace

int main() {
  // ---- CPU: the producer's in-flight bookkeeping (KafkaProducerActorImplSpec.scala:343-368, :692-705) -------------------------
  {
    surge::InFlightTracker t;
    bool ok = t.isAggregateStateCurrent("bar");
    t.addInFlight({{"bar", 101}});
    ok = ok && !t.isAggregateStateCurrent("bar") && t.isAggregateStateCurrent("foo");
    t.processedUpTo(100);
    ok = ok && !t.isAggregateStateCurrent("bar");
    t.processedUpTo(101);  // KTableProgressUpdate(LagInfo(101, 101))
    check(ok && t.isAggregateStateCurrent("bar"), "isAggregateStateCurrent follows KTable progress (in flight at 101 until the KTable reaches 101)");
    surge::InFlightTracker u;
    u.addInFlight({{"a", 5}, {"b", 6}, {"a", 9}, {"a", 7}});
    bool ok2 = u.inFlightForAggregate("a") == std::vector<int64_t>{9} && u.inFlightForAggregate("b") == std::vector<int64_t>{6};
    u.processedUpTo(8);
    check(ok2 && u.inFlightForAggregate("a") == std::vector<int64_t>{9} && u.inFlightForAggregate("b").empty(), "only the newest in-flight offset per key is kept");
  }
  // ---- CPU: KTableInitializationSupport's retry protocol against a fake S2 ------------------------------------------------------
  {
    using FakePublisher = surge::StatePublisher<FakeStore, Event>;
    using FakeActor = surge::GpuPersistentActor<State, Command, Event, FakeStore>;
    auto bl = businessLogic();
    std::vector<double> slept;
    auto store = std::make_shared<FakeStore>();
    const std::string three = jsonOf(State{"x", 3, 3});
    store->values["x"] = std::vector<uint8_t>(three.begin(), three.end());
    store->failures = 2;
    FakeActor a(bl, "x", std::make_shared<FakePublisher>(store), 0, {}, [&](double s) { slept.push_back(s); });
    const auto& st = a.getState();
    check(st && st->count == 3 && st->version == 3 && slept == std::vector<double>{2.0, 2.0} && a.initializationAttempts == 2,
          "two failed reads: retried after fetch-state-retry-interval (2 s each), then State(x,3,3)");
    slept.clear();
    auto pub = std::make_shared<FakePublisher>(store);
    pub->tracker.addInFlight({{"x", 7}});
    FakeActor b(bl, "x", pub, 0, {}, [&](double s) { slept.push_back(s); if (slept.size() == 3) pub->tracker.processedUpTo(7); });
    const auto& sb = b.getState();
    check(sb && sb->count == 3 && slept == std::vector<double>{0.5, 0.5, 0.5}, "not current in the KTable: 500 ms retries until the producer reports current");
    auto pub2 = std::make_shared<FakePublisher>(store);
    pub2->tracker.addInFlight({{"x", 7}});
    FakeActor c(bl, "x", pub2);
    bool threw = false;
    try { c.getState(); } catch (const surge::AggregateInitializationException&) { threw = true; }
    check(threw && c.initializationAttempts == 11, "never current: more than max-initialization-attempts (10) -> AggregateInitializationException");
    FakeActor d(bl, "nobody", std::make_shared<FakePublisher>(std::make_shared<FakeStore>()));
    check(!d.getState().has_value(), "a KTable miss is a valid initialisation: None");
  }

  // ---- GPU: the spec's scenarios, state of record in the GPU store -------------------------------------------------------------
  auto newContext = [&](const std::string& agg) {
    auto model = std::make_shared<CounterModel>();
    auto store = std::make_shared<Store>(model, std::make_shared<CounterFormat>(), 0);
    // TestContext.setupDefault: baseState = State(id, 3, 3) served by the KTable — here recovered from its three events
    store->restore({CountIncremented{agg, 1, 1}, CountIncremented{agg, 1, 2}, CountIncremented{agg, 1, 3}});
    return std::make_shared<Publisher>(store);
  };
  std::shared_ptr<Publisher> pub;
  try {
    pub = newContext("agg-1");
  } catch (const surge::ReplayException& e) {
    if (e.status() == SURGE_E_DEVICE) { std::printf("no GPU: %s\n%s\n", e.what(), fails ? "FAILED" : "CPU CHECKS PASS"); return fails ? 1 : 2; }
    std::printf("create failed: %s\n", e.what());
    return 1;
  }
  {
    Actor actor(businessLogic(), "agg-1", pub, 1);
    const auto& s0 = actor.getState();
    check(s0 && s0->count == 3 && s0->version == 3, "GetState initialises from the GPU store: Some(State(agg-1,3,3))");
    check(isState(actor.processMessage(Increment{"agg-1"}), "agg-1", 4, 4), "Increment on (3,3) -> ACKSuccess(State(agg-1,4,4))");
    const bool one = pub->published.size() == 1 && pub->published[0].size() == 2;
    const surge::PublishedRecord ev = one ? pub->published[0][0] : surge::PublishedRecord{};
    const surge::PublishedRecord st = one ? pub->published[0][1] : surge::PublishedRecord{};
    check(one && !ev.isState && ev.topic == "testEventsTopic" && ev.key == "agg-1:4", "one publish of two records: the event, keyed aggregateId:sequenceNumber ...");
    check(one && st.isState && st.topic == "testStateTopic" && st.partition == 1 && st.key == "agg-1" && text(st.value) == "{\"aggregateId\":\"agg-1\",\"count\":4,\"version\":4}",
          "... and the state record (state topic, assigned partition, key = aggregateId, value = writeState)");
    const bool before = !pub->tracker.isAggregateStateCurrent("agg-1");
    pub->ktableProgress();
    check(before && pub->tracker.isAggregateStateCurrent("agg-1") && pub->store().getAggregateBytes("agg-1") == st.value,
          "in flight until the KTable catches up; then the GPU store serves exactly the published bytes");
  }
  {
    pub = newContext("agg-1");
    Actor actor(businessLogic(), "agg-1", pub);
    bool ok = isState(actor.processMessage(DoNothing{"agg-1"}), "agg-1", 3, 3) && pub->published.empty();
    ok = ok && isState(actor.applyEvents({CountIncremented{"agg-1", 0, 3}}), "agg-1", 3, 3) && pub->published.empty();
    Actor stateOnly(businessLogic(true), "agg-1", pub);
    ok = ok && isState(stateOnly.processMessage(DoNothing{"agg-1"}), "agg-1", 3, 3) && pub->published.empty();
    check(ok, "nothing is published when nothing changed (DoNothing; ApplyEvents that leaves the state equal; publishStateOnly)");
  }
  {
    bool ok = true;
    for (const bool stateOnly : {false, true}) {
      pub = newContext("agg-1");
      Actor actor(businessLogic(stateOnly), "agg-1", pub);
      ok = ok && isState(actor.processMessage(Increment{"agg-1"}), "agg-1", 4, 4) && pub->published.size() == 1 &&
           pub->published[0].size() == (stateOnly ? 1u : 2u) && pub->published[0].back().isState;
    }
    check(ok, "publishStateOnly = false: event + state; true: the state record alone");
  }
  {
    pub = newContext("agg-1");
    Actor actor(businessLogic(), "agg-1", pub);
    bool ok = isError(actor.processMessage(FailCommandProcessing{"agg-1", "failed"}), "failed");
    ok = ok && isError(actor.processMessage(CreateExceptionThrowingEvent{"agg-1", "failed"}), "failed");
    ok = ok && isError(actor.applyEvents({ExceptionThrowingEvent{"agg-1", 1, "failed"}}), "failed") && pub->published.empty();
    ok = ok && isState(actor.processMessage(DoNothing{"agg-1"}), "agg-1", 3, 3);
    pub->ktableProgress();
    const std::optional<State> g = pub->store().getAggregate("agg-1");
    check(ok && g && g->count == 3 && g->version == 3, "domain exceptions become ACKError, publish nothing, and leave the actor (and the store) at (3,3)");
  }
  {
    pub = newContext("agg-1");
    Actor actor(businessLogic(), "agg-1", pub);
    bool ok = isState(actor.processMessage(Increment{"agg-1"}), "agg-1", 4, 4) && isState(actor.processMessage(Increment{"agg-1"}), "agg-1", 5, 5);
    check(ok, "commands one at a time: (4,4) then (5,5)");
    size_t n = pub->published.size();
    ok = isState(actor.processMessage(CreateNoOpEvent{"agg-1"}), "agg-1", 5, 5) && pub->published.size() == n + 1 && pub->published.back()[0].key == "agg-1:6";
    check(ok, "an event that does not change the state is still published");
    n = pub->published.size();
    ok = isState(actor.applyEvents({CountIncremented{"agg-1", 1, 6}}), "agg-1", 6, 6) && isState(actor.applyEvents({CountIncremented{"agg-1", 1, 7}}), "agg-1", 7, 7);
    ok = ok && pub->published.size() == n + 2 && pub->published[n].size() == 1 && pub->published[n][0].isState && pub->published[n + 1].size() == 1 &&
         pub->published[n + 1][0].isState;
    check(ok, "ApplyEvents publishes only state records: (6,6), (7,7)");
    pub->ktableProgress();
    ok = pub->store().getAggregateBytes("agg-1") == pub->published.back().back().value;
    Actor again(businessLogic(), "agg-1", pub);
    const auto& s = again.getState();
    check(ok && s && s->count == 7 && s->version == 7, "after the KTable caught up the GPU store equals the last state record; a new actor initialises to (7,7)");
  }
  {
    pub = newContext("someone-else");
    Actor actor(businessLogic(), "fresh", pub);
    bool ok = !actor.getState().has_value();
    ok = ok && isState(actor.processMessage(Increment{"fresh"}), "fresh", 1, 1) && isState(actor.processMessage(Increment{"fresh"}), "fresh", 2, 2) &&
         isState(actor.processMessage(Decrement{"fresh"}), "fresh", 1, 3);
    pub->ktableProgress();  // the aggregate did not exist at recovery: the resident state grows
    const std::optional<State> f = pub->store().getAggregate("fresh"), o = pub->store().getAggregate("someone-else");
    check(ok && f && f->count == 1 && f->version == 3 && o && o->count == 3 && o->version == 3,
          "multilanguage gateway sequence on a new aggregate: (1,1), (2,2), decrement -> (1,3); folded onto the grown GPU state");
  }
  std::printf("%s\n", fails ? "FAILED" : "ALL PASS");
  return fails ? 1 : 0;
}
