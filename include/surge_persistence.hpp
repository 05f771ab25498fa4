/* surge_persistence.hpp — C++17 mirror of the CALLERS of the fold, above surge_replay.hpp.
 *
 * The reference folds inside one actor per aggregate; what surrounds the fold decides what is published and when an
 * actor may trust the store.  This header restates that protocol synchronously (no actors) for a compiled-language host,
 * with the state of record in the GPU store of surge_replay.hpp:
 *
 *   AggregateCommandModel::toCore            CommandModels.scala:17-28 (modules/command-engine/scaladsl/.../command/):
 *                                            handle = processCommand, then events.foldLeft(state)(handleEvent);
 *                                            applyAsync = events.foldLeft(state)(handleEvent)
 *   SurgeContext / SurgeProcessingModel      modules/command-engine/core/src/main/scala/surge/internal/domain/
 *                                            AggregateProcessingModel.scala:17-64 (immutable result carrier)
 *   GpuPersistentActor::processMessage       modules/command-engine/core/src/main/scala/surge/internal/persistence/
 *                    / applyEvents           PersistentActor.scala:197-272: a command publishes its events AND the new state
 *                                            in one batch iff events.nonEmpty || records.nonEmpty || state changed (:212);
 *                                            ApplyEvents publishes ONLY the state and only when it changed (:255-257); an
 *                                            exception becomes ACKError and leaves the actor's state untouched
 *                                            (:227-229, :260-262); publishStateOnly drops the event records (:207-211)
 *   GpuPersistentActor::initialize           .../KTableInitializationSupport.scala:37-81: not current in the KTable ->
 *                                            retry after initialize-state-interval (500 ms); failed read -> retry after
 *                                            fetch-state-retry-interval (2 s); more than max-initialization-attempts (10)
 *                                            -> AggregateInitializationException (reference.conf of common :137-142)
 *   InFlightTracker                          .../internal/kafka/KafkaProducerActorImpl.scala:530-540, 684-705: a published
 *                                            state record is in flight until the KTable indexed its offset; an aggregate
 *                                            is current iff none of its records are
 *   StatePublisher                           KafkaProducerActor.publish seen from the actor: one call = one transaction;
 *                                            ktableProgress() = the KTable caught up: the batch's events are folded onto
 *                                            the GPU-resident state as ONE micro-batch (K3)
 *
 * Python twin (same protocol, same tests): surge_amd/persistence.py, tests/test_persistence.py.
 * Demo / test: examples/cpp_persistence_demo.cpp (the reference's PersistentActorSpec / KafkaProducerActorImplSpec /
 * MultilanguageGatewayServiceImplSpec expectations, state of record on the GPU).
 */
#ifndef SURGE_PERSISTENCE_HPP
#define SURGE_PERSISTENCE_HPP

#include <exception>
#include <functional>
#include <variant>

#include "surge_replay.hpp"

namespace surge {

// ---- R5: SurgeContext / SurgeProcessingModel --------------------------------------------------------------------------
struct KafkaTopic {
  std::string name;
};

template <class State, class Evt>
struct SurgeContext {
  std::optional<State> state;
  KafkaTopic defaultEventTopic;
  std::vector<std::pair<Evt, KafkaTopic>> events;  // (event, topic) in persist order
  std::vector<SerializedMessage> records;          // persistRecord: extra records of the same transaction
  bool isRejected = false;
  std::string rejection;

  SurgeContext persistEvent(const Evt& e) const {
    SurgeContext c = *this;
    c.events.emplace_back(e, defaultEventTopic);
    return c;
  }
  SurgeContext persistEvents(const std::vector<Evt>& es) const {
    SurgeContext c = *this;
    for (const Evt& e : es) c.events.emplace_back(e, defaultEventTopic);
    return c;
  }
  SurgeContext persistToTopic(const Evt& e, const KafkaTopic& topic) const {
    SurgeContext c = *this;
    c.events.emplace_back(e, topic);
    return c;
  }
  SurgeContext persistRecord(const SerializedMessage& r) const {
    SurgeContext c = *this;
    c.records.push_back(r);
    return c;
  }
  SurgeContext updateState(const std::optional<State>& s) const {
    SurgeContext c = *this;
    c.state = s;
    return c;
  }
  SurgeContext reject(const std::string& why) const {
    SurgeContext c = *this;
    c.isRejected = true;
    c.rejection = why;
    return c;
  }
};

template <class State, class Msg, class Evt>
struct SurgeProcessingModel {
  virtual ~SurgeProcessingModel() = default;
  virtual SurgeContext<State, Evt> handle(const SurgeContext<State, Evt>& ctx, const std::optional<State>& state, const Msg& msg) const = 0;
  virtual SurgeContext<State, Evt> applyAsync(const SurgeContext<State, Evt>& ctx, const std::optional<State>& state,
                                             const std::vector<Evt>& events) const = 0;
};

// ---- R3: trait AggregateCommandModel[Agg, Cmd, Evt] (processCommand joins the replay declaration) ----------------------
template <class Agg, class Cmd, class Evt>
struct AggregateCommandModel : ReplayableCommandModel<Agg, Evt> {
  // returns the events; throws where the reference returns Failure(e)
  virtual std::vector<Evt> processCommand(const std::optional<Agg>& aggregate, const Cmd& command) const = 0;
};

// AggregateCommandModel.toCore — CommandModels.scala:17-28
template <class Agg, class Cmd, class Evt>
class CommandModelCore : public SurgeProcessingModel<Agg, Cmd, Evt> {
 public:
  explicit CommandModelCore(std::shared_ptr<const AggregateCommandModel<Agg, Cmd, Evt>> model) : model_(std::move(model)) {}
  SurgeContext<Agg, Evt> handle(const SurgeContext<Agg, Evt>& ctx, const std::optional<Agg>& state, const Cmd& cmd) const override {
    const std::vector<Evt> events = model_->processCommand(state, cmd);
    return ctx.persistEvents(events).updateState(model_->applyEvents(state, events));  // events.foldLeft(state)(handleEvent)
  }
  SurgeContext<Agg, Evt> applyAsync(const SurgeContext<Agg, Evt>& ctx, const std::optional<Agg>& state,
                                     const std::vector<Evt>& events) const override {
    return ctx.updateState(model_->applyEvents(state, events));
  }

 private:
  std::shared_ptr<const AggregateCommandModel<Agg, Cmd, Evt>> model_;
};

template <class Evt>
struct SurgeEventWriteFormatting {
  virtual ~SurgeEventWriteFormatting() = default;
  virtual SerializedMessage writeEvent(const Evt& evt) const = 0;
};

// SurgeCommandBusinessLogic — the bundle an engine is built from (commondsl/SurgeGenericBusinessLogicTrait.scala)
template <class Agg, class Cmd, class Evt>
struct SurgeCommandBusinessLogic {
  std::string aggregateName;
  KafkaTopic stateTopic, eventsTopic;
  std::shared_ptr<const AggregateCommandModel<Agg, Cmd, Evt>> commandModel;
  std::shared_ptr<const SurgeAggregateReadFormatting<Agg>> aggregateReadFormatting;
  std::shared_ptr<const SurgeAggregateWriteFormatting<Agg>> aggregateWriteFormatting;
  std::shared_ptr<const SurgeEventWriteFormatting<Evt>> eventWriteFormatting;
  bool publishStateOnly = false;
};

// ---- R11: the producer's in-flight bookkeeping for one state-topic partition ------------------------------------------
class InFlightTracker {
 public:
  // (key, offset) of just-published state records; only the newest offset per key is kept (:692-705)
  void addInFlight(const std::vector<std::pair<std::string, int64_t>>& records) {
    for (const auto& r : records) {
      auto it = inFlight_.find(r.first);
      if (it == inFlight_.end() || r.second > it->second) inFlight_[r.first] = r.second;
    }
  }
  // KTableProgressUpdate(LagInfo(currentOffsetPosition = o, ...)): every record with offset <= o is indexed (:684-698)
  void processedUpTo(int64_t ktableCurrentOffset) {
    for (auto it = inFlight_.begin(); it != inFlight_.end();) it = it->second <= ktableCurrentOffset ? inFlight_.erase(it) : std::next(it);
  }
  std::vector<int64_t> inFlightForAggregate(const std::string& aggregateId) const {
    const auto it = inFlight_.find(aggregateId);
    return it == inFlight_.end() ? std::vector<int64_t>{} : std::vector<int64_t>{it->second};
  }
  // IsAggregateStateCurrent -> noRecordsInFlight (:530-534)
  bool isAggregateStateCurrent(const std::string& aggregateId) const { return inFlight_.find(aggregateId) == inFlight_.end(); }

 private:
  std::map<std::string, int64_t> inFlight_;
};

// what a publish call carries: SurgeModel.serializeState / serializeEvents (SurgeModel.scala:37-65)
struct PublishedRecord {
  bool isState = false;
  std::string topic;
  int32_t partition = -1;                        // the state record names its partition explicitly (:57-65)
  std::string key;
  std::optional<std::vector<uint8_t>> value;     // nullopt = tombstone (state is None)
  std::map<std::string, std::string> headers;
};

class AggregateStateNotCurrentInKTableException : public std::runtime_error {
 public:
  explicit AggregateStateNotCurrentInKTableException(const std::string& id) : std::runtime_error("aggregate " + id + " is not current in the KTable") {}
};

struct RetryConfig {  // surge.aggregate-actor.* (reference.conf of common :137-142)
  double initializeStateIntervalS = 0.5;
  double fetchStateRetryIntervalS = 2.0;
  int maxInitializationAttempts = 10;
};

// Where a batch of records goes.  `Store` needs getAggregateBytes(id) and applyEvents(vector<Evt>) — AggregateStateStore
// of surge_replay.hpp, or a fake in a test of the retry logic.
template <class Store, class Evt>
class StatePublisher {
 public:
  explicit StatePublisher(std::shared_ptr<Store> store) : store_(std::move(store)) {}
  void publish(const std::string& /*aggregateId*/, std::vector<PublishedRecord> records, const std::vector<Evt>& events) {
    std::vector<std::pair<std::string, int64_t>> flight;
    for (const PublishedRecord& r : records)
      if (r.isState) flight.emplace_back(r.key, nextOffset_++);
    tracker.addInFlight(flight);
    published.push_back(std::move(records));
    pending_.insert(pending_.end(), events.begin(), events.end());
  }
  // the KTable catches up: pending events fold onto the GPU store (one micro-batch for all aggregates), offsets retire
  void ktableProgress() {
    if (!pending_.empty()) {
      store_->applyEvents(pending_);
      pending_.clear();
    }
    tracker.processedUpTo(nextOffset_ - 1);
  }
  Store& store() const { return *store_; }

  InFlightTracker tracker;
  std::vector<std::vector<PublishedRecord>> published;  // one entry per publish call (= one Kafka transaction)

 private:
  std::shared_ptr<Store> store_;
  std::vector<Evt> pending_;
  int64_t nextOffset_ = 0;
};

// ---- R10: one aggregate's PersistentActor, synchronous ---------------------------------------------------------------------
template <class Agg>
struct ACKSuccess {  // PersistentActor.ACKSuccess(aggregateState: Option[S]) (:45-47)
  std::optional<Agg> aggregateState;
};
struct ACKError {    // PersistentActor.ACKError(exception) (:48-50)
  std::string what;
  std::exception_ptr exception;
};
template <class Agg>
using Ack = std::variant<ACKSuccess<Agg>, ACKError>;

template <class Agg, class Cmd, class Evt, class Store>
class GpuPersistentActor {
 public:
  GpuPersistentActor(std::shared_ptr<const SurgeCommandBusinessLogic<Agg, Cmd, Evt>> businessLogic, std::string aggregateId,
                     std::shared_ptr<StatePublisher<Store, Evt>> publisher, int32_t assignedPartition = 0, RetryConfig retry = {},
                     std::function<void(double)> sleep = [](double) {})
      : bl_(std::move(businessLogic)), core_(bl_->commandModel), id_(std::move(aggregateId)), pub_(std::move(publisher)),
        partition_(assignedPartition), retry_(retry), sleep_(std::move(sleep)) {}

  // KTableInitializationSupport.initializeState / fetchState
  void initialize() {
    std::string cause = "never tried";
    int attempts = 0;
    for (;;) {
      if (attempts > retry_.maxInitializationAttempts) {
        initializationAttempts = attempts;
        throw AggregateInitializationException("Aggregate " + id_ + " could not be initialized: " + cause);
      }
      if (!pub_->tracker.isAggregateStateCurrent(id_)) {
        cause = AggregateStateNotCurrentInKTableException(id_).what();
        sleep_(retry_.initializeStateIntervalS);
        ++attempts;
        continue;
      }
      try {
        const std::optional<std::vector<uint8_t>> bytes = pub_->store().getAggregateBytes(id_);  // seam S2
        state_ = bytes ? bl_->aggregateReadFormatting->readState(*bytes) : std::nullopt;
        initialized_ = true;
        initializationAttempts = attempts;
        return;
      } catch (const AggregateInitializationException&) {
        throw;  // the replay of this aggregate threw: not a transient read failure
      } catch (const std::exception& e) {  // a failed read -> fetchState's recover -> retry
        cause = e.what();
        sleep_(retry_.fetchStateRetryIntervalS);
        ++attempts;
      }
    }
  }

  // PersistentActor.handle (:197-232)
  Ack<Agg> processMessage(const Cmd& message) {
    if (!initialized_) initialize();
    try {
      SurgeContext<Agg, Evt> start;
      start.state = state_;
      start.defaultEventTopic = bl_->eventsTopic;
      const SurgeContext<Agg, Evt> ctx = core_.handle(start, state_, message);
      if (ctx.isRejected) return ACKError{"rejected: " + ctx.rejection, nullptr};
      std::vector<Evt> events;
      for (const auto& et : ctx.events) events.push_back(et.first);
      const bool isSomethingNew = !events.empty() || !ctx.records.empty() || !sameState(state_, ctx.state);
      std::vector<PublishedRecord> records;
      if (!bl_->publishStateOnly)
        for (const auto& et : ctx.events) records.push_back(serializeEvent(et.first, et.second));
      for (const SerializedMessage& r : ctx.records) records.push_back(PublishedRecord{false, bl_->eventsTopic.name, -1, r.key, r.value, r.headers});
      records.push_back(serializeState(ctx.state));
      if (isSomethingNew) pub_->publish(id_, std::move(records), events);
      state_ = ctx.state;
      return ACKSuccess<Agg>{ctx.state};
    } catch (const std::exception& e) {  // .recover { case e => ACKError(e) } — the actor's state is untouched
      return ACKError{e.what(), std::current_exception()};
    }
  }

  // PersistentActor.doApplyEvent (:245-264): state-only publish, and only when the state changed
  Ack<Agg> applyEvents(const std::vector<Evt>& events) {
    if (!initialized_) initialize();
    try {
      SurgeContext<Agg, Evt> start;
      start.state = state_;
      start.defaultEventTopic = bl_->eventsTopic;
      const SurgeContext<Agg, Evt> ctx = core_.applyAsync(start, state_, events);
      if (!sameState(state_, ctx.state)) pub_->publish(id_, {serializeState(ctx.state)}, events);
      state_ = ctx.state;
      return ACKSuccess<Agg>{ctx.state};
    } catch (const std::exception& e) {
      return ACKError{e.what(), std::current_exception()};
    }
  }

  // PersistentActor.GetState -> StateResponse (:52-53)
  const std::optional<Agg>& getState() {
    if (!initialized_) initialize();
    return state_;
  }

  int initializationAttempts = 0;

 private:
  // the reference compares case classes; here: the bytes writeState produces (what a reader of the topic could tell apart)
  bool sameState(const std::optional<Agg>& a, const std::optional<Agg>& b) const {
    if (a.has_value() != b.has_value()) return false;
    return !a || bl_->aggregateWriteFormatting->writeState(*a).value == bl_->aggregateWriteFormatting->writeState(*b).value;
  }
  PublishedRecord serializeState(const std::optional<Agg>& s) const {
    if (!s) return PublishedRecord{true, bl_->stateTopic.name, partition_, id_, std::nullopt, {}};
    SerializedAggregate ser = bl_->aggregateWriteFormatting->writeState(*s);
    return PublishedRecord{true, bl_->stateTopic.name, partition_, id_, std::move(ser.value), std::move(ser.headers)};
  }
  PublishedRecord serializeEvent(const Evt& e, const KafkaTopic& topic) const {
    SerializedMessage m = bl_->eventWriteFormatting->writeEvent(e);
    return PublishedRecord{false, topic.name.empty() ? bl_->eventsTopic.name : topic.name, -1, std::move(m.key), std::move(m.value), std::move(m.headers)};
  }

  std::shared_ptr<const SurgeCommandBusinessLogic<Agg, Cmd, Evt>> bl_;
  CommandModelCore<Agg, Cmd, Evt> core_;
  std::string id_;
  std::shared_ptr<StatePublisher<Store, Evt>> pub_;
  int32_t partition_;
  RetryConfig retry_;
  std::function<void(double)> sleep_;
  std::optional<Agg> state_;
  bool initialized_ = false;
};

}  // namespace surge

#endif  // SURGE_PERSISTENCE_HPP
