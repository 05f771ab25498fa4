"""The reference's own sample models, restated on the host mirror (used by tests and examples).

* Counter — ``modules/command-engine/scaladsl/src/test/scala/surge/scaladsl/TestBoundedContext.scala``
  (State :15, commands :18-39, events :49-69, handleEvent :77-89, processCommand :91-106,
  formats :122-134).
* BankAccount — ``modules/surge-docs/src/test/scala/docs/command/BankAccountCommandModel.scala``
  (aggregate :19, commands :23-28, events :32-48, processCommand :53-79, handleEvent :81-86).

``handle_event`` below is the literal case analysis of the Scala code (JVM ``Int`` wrap included);
the ``event_algebra`` beside it is what the GPU replays.  The two are checked against each other
in ``tests/test_host_models.py``.
"""
from __future__ import annotations

import json
import uuid
from dataclasses import dataclass, replace
from typing import Optional, Sequence, Union

import numpy as np

from surge_amd.command import ReplayableCommandModel, SurgeCommandBusinessLogic
from surge_amd.core import (
    KafkaTopic,
    SerializedAggregate,
    SerializedMessage,
    SurgeAggregateFormatting,
    SurgeEventReadFormatting,
    SurgeEventWriteFormatting,
)
from surge_amd.schema import (
    CLS_CREATE,
    CLS_MATERIALIZE,
    CLS_REQUIRE,
    D_BALANCE_SET,
    D_COUNT_ADD,
    D_COUNT_SUB,
    D_POISON,
    D_VERSION_SET,
    STATE_DTYPE,
    STATE_PRESENT,
    EventAlgebra,
)


_JSON_ESC = {'"': '\\"', "\\": "\\\\", "\b": "\\b", "\f": "\\f", "\n": "\\n", "\r": "\\r", "\t": "\\t"}


def jackson_quote(s: str) -> str:
    """JSON string with Jackson's default escaping (what play-json's ``Json.stringify`` emits): the seven
    short escapes, other controls as ``\\u00XX`` with upper-case hex, everything else verbatim (UTF-8)."""
    out = ['"']
    for ch in s:
        if ch in _JSON_ESC:
            out.append(_JSON_ESC[ch])
        elif ord(ch) < 0x20:
            out.append("\\u%04X" % ord(ch))
        else:
            out.append(ch)
    out.append('"')
    return "".join(out)


def _i32(x: int) -> int:
    """JVM ``Int`` arithmetic: wrap to 32-bit two's complement."""
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


# ======================================================================================
# Counter (TestBoundedContext)
# ======================================================================================
@dataclass(frozen=True)
class State:
    aggregateId: str
    count: int
    version: int


@dataclass(frozen=True)
class Increment:
    aggregateId: str


@dataclass(frozen=True)
class Decrement:
    aggregateId: str


@dataclass(frozen=True)
class DoNothing:
    aggregateId: str


@dataclass(frozen=True)
class CreateNoOpEvent:
    aggregateId: str


@dataclass(frozen=True)
class FailCommandProcessing:
    aggregateId: str
    withError: Exception


@dataclass(frozen=True)
class CreateExceptionThrowingEvent:
    aggregateId: str
    throwable: Exception


@dataclass(frozen=True)
class CountIncremented:
    aggregateId: str
    incrementBy: int
    sequenceNumber: int
    eventName = "countIncremented"


@dataclass(frozen=True)
class CountDecremented:
    aggregateId: str
    decrementBy: int
    sequenceNumber: int
    eventName = "countDecremented"


@dataclass(frozen=True)
class NoOpEvent:
    aggregateId: str
    sequenceNumber: int
    eventName = "no-op"


@dataclass(frozen=True)
class ExceptionThrowingEvent:
    aggregateId: str
    sequenceNumber: int
    throwable: Exception
    eventName = "exception-throwing"


CounterEvent = Union[CountIncremented, CountDecremented, NoOpEvent, ExceptionThrowingEvent]

#: event type codes of the Counter model's own (4-type) algebra
CT_NOOP, CT_INC, CT_DEC, CT_THROW = 0, 1, 2, 3

COUNTER_ALGEBRA = EventAlgebra(
    desc=(
        CLS_MATERIALIZE,                                   # NoOpEvent           :85
        CLS_MATERIALIZE | D_COUNT_ADD | D_VERSION_SET,     # CountIncremented    :81-82
        CLS_MATERIALIZE | D_COUNT_SUB | D_VERSION_SET,     # CountDecremented    :83-84
        D_POISON,                                          # ExceptionThrowingEvent :86
    ),
    names=("no-op", "countIncremented", "countDecremented", "exception-throwing"),
)


class CounterCommandModel(ReplayableCommandModel[State, object, CounterEvent]):
    # TestBoundedContext.scala:77-89
    def handle_event(self, agg: Optional[State], evt: CounterEvent) -> Optional[State]:
        current = agg if agg is not None else State(evt.aggregateId, 0, 0)
        if isinstance(evt, CountIncremented):
            new_state = replace(current, count=_i32(current.count + evt.incrementBy), version=evt.sequenceNumber)
        elif isinstance(evt, CountDecremented):
            new_state = replace(current, count=_i32(current.count - evt.decrementBy), version=evt.sequenceNumber)
        elif isinstance(evt, NoOpEvent):
            new_state = current
        elif isinstance(evt, ExceptionThrowingEvent):
            raise evt.throwable
        else:
            raise TypeError(f"MatchError: {evt!r}")
        return new_state

    # TestBoundedContext.scala:91-106
    def process_command(self, agg: Optional[State], cmd) -> Sequence[CounterEvent]:
        new_sequence_number = (agg.version if agg is not None else 0) + 1
        if isinstance(cmd, Increment):
            return [CountIncremented(cmd.aggregateId, 1, new_sequence_number)]
        if isinstance(cmd, Decrement):
            return [CountDecremented(cmd.aggregateId, 1, new_sequence_number)]
        if isinstance(cmd, CreateNoOpEvent):
            return [NoOpEvent(cmd.aggregateId, new_sequence_number)]
        if isinstance(cmd, DoNothing):
            return []
        if isinstance(cmd, FailCommandProcessing):
            raise cmd.withError
        if isinstance(cmd, CreateExceptionThrowingEvent):
            return [ExceptionThrowingEvent(cmd.aggregateId, new_sequence_number, cmd.throwable)]
        raise RuntimeError("Received unexpected message in command handler! This should not happen and indicates a bad test")

    # ---- additive: the replay declaration -------------------------------------------------
    def event_algebra(self) -> EventAlgebra:
        return COUNTER_ALGEBRA

    def encode_event(self, event: CounterEvent):
        if isinstance(event, CountIncremented):
            return CT_INC, event.sequenceNumber, event.incrementBy, None
        if isinstance(event, CountDecremented):
            return CT_DEC, event.sequenceNumber, event.decrementBy, None
        if isinstance(event, NoOpEvent):
            return CT_NOOP, event.sequenceNumber, 0, None
        if isinstance(event, ExceptionThrowingEvent):
            return CT_THROW, event.sequenceNumber, 0, None
        raise TypeError(f"not a Counter event: {event!r}")

    def event_json_template(self):
        """The JSON ``CounterEventFormat.write_event`` produces (``Json.toJson(evt)`` over the sealed ``BaseTestEvent`` format,
        TestBoundedContext.scala:42-49): a ``_type`` discriminator, ``sequenceNumber`` and the increment / decrement."""
        from surge_amd.ingest import ARG_I32, ARG_NONE, EventJsonTemplate

        return EventJsonTemplate("_type", [("countIncremented", CT_INC, "sequenceNumber", "incrementBy", ARG_I32),
                                           ("countDecremented", CT_DEC, "sequenceNumber", "decrementBy", ARG_I32),
                                           ("no-op", CT_NOOP, "sequenceNumber", "", ARG_NONE)])

    def aggregate_id_of(self, event: CounterEvent) -> str:
        return event.aggregateId

    def state_from_fixed(self, aggregate_id: str, fixed) -> State:
        return State(aggregate_id, int(fixed["count"]), int(fixed["version"]))

    def state_to_fixed(self, aggregate: State) -> np.ndarray:
        s = np.zeros(1, dtype=STATE_DTYPE)
        a = self.event_algebra()
        s["min_arg"], s["max_arg"] = a.default_min_arg, a.default_max_arg
        s["count"], s["version"], s["flags"] = aggregate.count, aggregate.version, STATE_PRESENT
        return s


class CounterAggregateFormat(SurgeAggregateFormatting[State]):
    """``Json.toJson(agg).toString().getBytes()`` / ``Json.parse(bytes).asOpt[State]`` (:126-134).

    play-json's ``Json.format`` macro writes the case-class fields in declaration order, compact.
    """

    def write_state(self, agg: State) -> SerializedAggregate:
        text = '{"aggregateId":%s,"count":%d,"version":%d}' % (jackson_quote(agg.aggregateId), agg.count, agg.version)
        return SerializedAggregate(text.encode("utf-8"))

    def read_state(self, data: bytes) -> Optional[State]:
        try:
            o = json.loads(data)
            return State(str(o["aggregateId"]), int(o["count"]), int(o["version"]))
        except Exception:
            return None  # asOpt


class CounterEventFormat(SurgeEventWriteFormatting[CounterEvent], SurgeEventReadFormatting[CounterEvent]):
    """Key ``"<aggregateId>:<sequenceNumber>"`` + JSON value (:122-124).  ``read_event`` is additive."""

    def write_event(self, evt: CounterEvent) -> SerializedMessage:
        body = {"aggregateId": evt.aggregateId}
        if isinstance(evt, CountIncremented):
            body["incrementBy"] = evt.incrementBy
        elif isinstance(evt, CountDecremented):
            body["decrementBy"] = evt.decrementBy
        body["sequenceNumber"] = evt.sequenceNumber
        body["_type"] = evt.eventName  # discriminator of the sealed-trait Format (:48)
        return SerializedMessage(
            f"{evt.aggregateId}:{evt.sequenceNumber}", json.dumps(body, separators=(",", ":")).encode("utf-8")
        )

    def read_event(self, msg: SerializedMessage) -> CounterEvent:
        o = json.loads(msg.value)
        t = o.get("_type")
        if t == "countIncremented":
            return CountIncremented(o["aggregateId"], int(o["incrementBy"]), int(o["sequenceNumber"]))
        if t == "countDecremented":
            return CountDecremented(o["aggregateId"], int(o["decrementBy"]), int(o["sequenceNumber"]))
        if t == "no-op":
            return NoOpEvent(o["aggregateId"], int(o["sequenceNumber"]))
        raise ValueError(f"unreadable event {msg.key}")


class CounterBusinessLogic(SurgeCommandBusinessLogic[State, object, CounterEvent]):
    """``TestBoundedContext.businessLogic`` (:136-155)."""

    aggregate_name = "CounterAggregate"
    state_topic = KafkaTopic("testStateTopic")
    events_topic = KafkaTopic("testEventsTopic")

    def __init__(self):
        self._model = CounterCommandModel()
        self._agg_format = CounterAggregateFormat()
        self._evt_format = CounterEventFormat()

    def command_model(self):
        return self._model

    def aggregate_read_formatting(self):
        return self._agg_format

    def aggregate_write_formatting(self):
        return self._agg_format

    def event_write_formatting(self):
        return self._evt_format


# ======================================================================================
# BankAccount (surge-docs sample)
# ======================================================================================
@dataclass(frozen=True)
class BankAccount:
    accountNumber: uuid.UUID
    accountOwner: str
    securityCode: str
    balance: float


@dataclass(frozen=True)
class CreateAccount:
    accountNumber: uuid.UUID
    accountOwner: str
    securityCode: str
    initialBalance: float


@dataclass(frozen=True)
class CreditAccount:
    accountNumber: uuid.UUID
    amount: float


@dataclass(frozen=True)
class DebitAccount:
    accountNumber: uuid.UUID
    amount: float


@dataclass(frozen=True)
class BankAccountCreated:
    accountNumber: uuid.UUID
    accountOwner: str
    securityCode: str
    balance: float


@dataclass(frozen=True)
class BankAccountUpdated:
    accountNumber: uuid.UUID
    newBalance: float


class AccountDoesNotExistException(Exception):
    pass


class InsufficientFundsException(Exception):
    pass


BA_CREATED, BA_UPDATED = 0, 1
BANK_ACCOUNT_ALGEBRA = EventAlgebra(
    desc=(
        CLS_CREATE | D_BALANCE_SET,    # BankAccountCreated :83
        CLS_REQUIRE | D_BALANCE_SET,   # BankAccountUpdated :84
    ),
    names=("BankAccountCreated", "BankAccountUpdated"),
)


class BankAccountCommandModel(ReplayableCommandModel[BankAccount, object, object]):
    """The non-numeric fields (owner, security code) ride in a host-side side table keyed by the
    account number: they are set once by ``BankAccountCreated`` and never folded."""

    def __init__(self):
        self.static_fields = {}

    # BankAccountCommandModel.scala:53-79
    def process_command(self, aggregate: Optional[BankAccount], command):
        if isinstance(command, CreateAccount):
            if aggregate is not None:
                return []
            return [BankAccountCreated(command.accountNumber, command.accountOwner, command.securityCode, command.initialBalance)]
        if isinstance(command, CreditAccount):
            if aggregate is None:
                raise AccountDoesNotExistException(command.accountNumber)
            return [BankAccountUpdated(aggregate.accountNumber, aggregate.balance + command.amount)]
        if isinstance(command, DebitAccount):
            if aggregate is None:
                raise AccountDoesNotExistException(command.accountNumber)
            if aggregate.balance >= command.amount:
                return [BankAccountUpdated(aggregate.accountNumber, aggregate.balance - command.amount)]
            raise InsufficientFundsException(aggregate.accountNumber)
        raise TypeError(f"MatchError: {command!r}")

    # BankAccountCommandModel.scala:81-86
    def handle_event(self, aggregate: Optional[BankAccount], event) -> Optional[BankAccount]:
        if isinstance(event, BankAccountCreated):
            return BankAccount(event.accountNumber, event.accountOwner, event.securityCode, event.balance)
        if isinstance(event, BankAccountUpdated):
            return None if aggregate is None else replace(aggregate, balance=event.newBalance)
        raise TypeError(f"MatchError: {event!r}")

    def event_algebra(self) -> EventAlgebra:
        return BANK_ACCOUNT_ALGEBRA

    def encode_event(self, event):
        if isinstance(event, BankAccountCreated):
            self.static_fields[str(event.accountNumber)] = (event.accountOwner, event.securityCode)
            return BA_CREATED, 0, None, event.balance
        if isinstance(event, BankAccountUpdated):
            return BA_UPDATED, 0, None, event.newBalance
        raise TypeError(f"not a BankAccount event: {event!r}")

    def event_json_template(self):
        """``Json.toJson(evt)(Json.format[BankAccountEvent])`` (BankAccountSurgeModel.scala:30-32): play-json's sealed-family
        format names the case class in ``_type`` (fully qualified by default); the balances are JSON numbers."""
        from surge_amd.ingest import ARG_F64, EventJsonTemplate

        return EventJsonTemplate("_type", [("docs.command.BankAccountCreated", BA_CREATED, "", "balance", ARG_F64),
                                           ("docs.command.BankAccountUpdated", BA_UPDATED, "", "newBalance", ARG_F64)])

    def aggregate_id_of(self, event) -> str:
        return str(event.accountNumber)

    def state_from_fixed(self, aggregate_id: str, fixed) -> BankAccount:
        owner, code = self.static_fields.get(aggregate_id, ("", ""))
        return BankAccount(uuid.UUID(aggregate_id), owner, code, float(fixed["balance"]))

    def state_to_fixed(self, aggregate: BankAccount) -> np.ndarray:
        s = np.zeros(1, dtype=STATE_DTYPE)
        a = self.event_algebra()
        s["min_arg"], s["max_arg"] = a.default_min_arg, a.default_max_arg
        s["balance"], s["flags"] = aggregate.balance, STATE_PRESENT
        self.static_fields[str(aggregate.accountNumber)] = (aggregate.accountOwner, aggregate.securityCode)
        return s


class BankAccountFormat(SurgeAggregateFormatting[BankAccount]):
    """``BankAccountSurgeModel`` formats (``.../docs/command/BankAccountSurgeModel.scala:22-32``).

    ``balance`` is written as play-json 2.9.2 writes a Scala ``Double`` (shortest round-trip digits through
    ``BigDecimal``: ``100.0`` -> ``100``; ``surge_amd/csrc/f64_text.h``), the same conversion the GPU encoder runs.
    """

    def write_state(self, agg: BankAccount) -> SerializedAggregate:
        from surge_amd.encode import play_json_double

        q = lambda v: json.dumps(v, ensure_ascii=False)  # noqa: E731  Jackson's default string escaping
        text = (f'{{"accountNumber":{q(str(agg.accountNumber))},"accountOwner":{q(agg.accountOwner)},'
                f'"securityCode":{q(agg.securityCode)},"balance":{play_json_double(agg.balance)}}}')
        return SerializedAggregate(text.encode("utf-8"), {"aggregate_id": str(agg.accountNumber)})

    def read_state(self, data: bytes) -> Optional[BankAccount]:
        try:
            o = json.loads(data)
            return BankAccount(uuid.UUID(o["accountNumber"]), o["accountOwner"], o["securityCode"], float(o["balance"]))
        except Exception:
            return None


class BankAccountEventFormat:
    """``BankAccountSurgeModel.eventWriteFormatting`` (``.../docs/command/BankAccountSurgeModel.scala:30-32``):
    ``SerializedMessage(evt.accountNumber.toString, Json.toJson(evt)(Json.format[BankAccountEvent]))`` — the record key is
    the account's UUID alone (no ``:<seq>``: these events carry no sequence number), the value the case class's fields in
    declaration order followed by play-json's sealed-family discriminator ``_type`` (the fully qualified class name), the
    balances written as play-json writes a ``Double``."""

    def write_event(self, evt) -> SerializedMessage:
        from surge_amd.encode import play_json_double

        q = lambda v: json.dumps(v, ensure_ascii=False)  # noqa: E731
        if isinstance(evt, BankAccountCreated):
            text = (f'{{"accountNumber":{q(str(evt.accountNumber))},"accountOwner":{q(evt.accountOwner)},"securityCode":{q(evt.securityCode)},'
                    f'"balance":{play_json_double(evt.balance)},"_type":"docs.command.BankAccountCreated"}}')
        elif isinstance(evt, BankAccountUpdated):
            text = (f'{{"accountNumber":{q(str(evt.accountNumber))},"newBalance":{play_json_double(evt.newBalance)},'
                    f'"_type":"docs.command.BankAccountUpdated"}}')
        else:
            raise TypeError(f"not a BankAccount event: {evt!r}")
        return SerializedMessage(str(evt.accountNumber), text.encode("utf-8"))


# ======================================================================================================
# Multilanguage Scala SDK sample: the second copy of the fold, behind the gRPC bridge.
#   CQRSModel.applyEvents = e.foldLeft(s)(eventHandler)
#       modules/multilanguage-scala-sdk/src/main/scala/com/ukg/surge/multilanguage/scalasdk/Model.scala:9-19
#   sample model BankAccount(balance: Int) / MoneyDeposited(amount: Int) / DepositMoney(amount: Int)
#       modules/multilanguage-scala-sdk-sample/src/main/scala/com/ukg/surge/multilanguage/scalasdk/sample/Main.scala:19-37
#   stored state = protobuf State{aggregateId, payload = json4s write(state)} (GenericSurgeCommandBusinessLogic.scala:36-39,
#       Main.scala:41-60)
# ======================================================================================================
@dataclass(frozen=True)
class SdkBankAccount:
    balance: int


@dataclass(frozen=True)
class MoneyDeposited:
    amount: int


@dataclass(frozen=True)
class DepositMoney:
    amount: int


@dataclass(frozen=True)
class SdkEvent:
    """The SDK's events carry no aggregate id of their own: the bridge passes it beside the payload
    (``Event{aggregateId, payload}``, multilanguage-protocol.proto:17-20)."""

    aggregateId: str
    payload: MoneyDeposited


class CQRSModel:
    """``CQRSModel[S, E, C](eventHandler, commandHandler)`` — Model.scala:9-19."""

    def __init__(self, event_handler, command_handler):
        self.event_handler = event_handler
        self.command_handler = command_handler

    def apply_events(self, s, events):
        for e in events:  # e.foldLeft(s)((s, e) => eventHandler(s, e)) — Model.scala:11-13
            s = self.event_handler(s, e)
        return s

    def execute_command(self, s, c):
        """``Left(msg)`` -> ``("Left", msg)``; ``Right((events, newState))`` -> ``("Right", (events, new_state))``."""
        tag, val = self.command_handler(s, c)
        if tag == "Left":
            return tag, val
        return "Right", (val, self.apply_events(s, val))


def sdk_sample_event_handler(agg: Optional[SdkBankAccount], evt: MoneyDeposited) -> Optional[SdkBankAccount]:
    # Main.scala:25-30
    if agg is None:
        return SdkBankAccount(_i32(evt.amount))
    return SdkBankAccount(_i32(agg.balance + evt.amount))


def sdk_sample_command_handler(agg: Optional[SdkBankAccount], cmd: DepositMoney):
    # Main.scala:32-38
    if cmd.amount >= 0:
        return "Right", [MoneyDeposited(cmd.amount)]
    return "Left", "Amount cannot be < 0"


SDK_SAMPLE_MODEL = CQRSModel(sdk_sample_event_handler, sdk_sample_command_handler)

SDK_DEPOSITED = 0
SDK_SAMPLE_ALGEBRA = EventAlgebra(
    desc=(CLS_MATERIALIZE | D_COUNT_ADD,),  # None => BankAccount(0 + amount); Some(b) => BankAccount(b + amount)
    names=("MoneyDeposited",),
)


class SdkSampleCommandModel(ReplayableCommandModel[SdkBankAccount, DepositMoney, SdkEvent]):
    """The sample's ``CQRSModel`` behind the replayable-model interface; ``balance`` lives in the ``count`` field."""

    def process_command(self, aggregate, command):
        tag, val = SDK_SAMPLE_MODEL.command_handler(aggregate, command)
        if tag == "Left":
            raise ValueError(val)
        return val

    def handle_event(self, aggregate: Optional[SdkBankAccount], event: SdkEvent) -> Optional[SdkBankAccount]:
        return SDK_SAMPLE_MODEL.event_handler(aggregate, event.payload)

    def event_algebra(self) -> EventAlgebra:
        return SDK_SAMPLE_ALGEBRA

    def encode_event(self, event: SdkEvent):
        return SDK_DEPOSITED, 0, event.payload.amount, None

    def aggregate_id_of(self, event: SdkEvent) -> str:
        return event.aggregateId

    def state_from_fixed(self, aggregate_id: str, fixed) -> SdkBankAccount:
        return SdkBankAccount(int(fixed["count"]))

    def state_to_fixed(self, aggregate: SdkBankAccount) -> np.ndarray:
        s = np.zeros(1, dtype=STATE_DTYPE)
        a = self.event_algebra()
        s["min_arg"], s["max_arg"] = a.default_min_arg, a.default_max_arg
        s["count"], s["flags"] = aggregate.balance, STATE_PRESENT
        return s


def protobuf_varint(v: int) -> bytes:
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def sdk_sample_state_bytes(aggregate_id: str, state: SdkBankAccount) -> bytes:
    """``protobuf.State(aggregateId, ByteString(write(state))).toByteArray``: json4s' compact text in field 2, the
    id in field 1; proto3 leaves an empty field out."""
    payload = ('{"balance":%d}' % state.balance).encode("ascii")
    key = aggregate_id.encode("utf-8")
    out = b""
    if key:
        out += b"\x0a" + protobuf_varint(len(key)) + key
    return out + b"\x12" + protobuf_varint(len(payload)) + payload
