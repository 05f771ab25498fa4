/*
 * surge_fold_oracle.c — CPU restatement of the reference's aggregate fold.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under surge_amd/ may include, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / reported baseline.
 *
 * PARITY STATUS: the reference is JVM-only and cannot be built or run in this
 * image (no JVM, sbt or jars), and it ships no golden files.  The
 * fold below is pinned against the explicit expected values in the reference's
 * own specs (tests/test_oracle_kat.py, SURVEY §8c KAT 1-7).  Two third-party
 * behaviours are "parity unpinned" (no reference test fixes their bytes):
 *   - scala.util.hashing.MurmurHash3.stringHash (scala-library 2.13.8) used by
 *     KafkaPartitioner.scala:8 — restated from the published algorithm; its mix / mixLast /
 *     finalize primitives ARE pinned, on the published MurmurHash3_x86_32 verification vectors
 *     (oracle_murmur3_x86_32 below is assembled from the same primitives) and against scikit-learn's
 *     bundled reference MurmurHash3_x86_32 on random inputs; what remains unpinned is stringHash's
 *     framing (seed, char-pair order, char count) — tools/MurmurPin.scala;
 *   - play-json 2.9.2 number/text formatting used by TestBoundedContext.scala:127-133 and (Doubles)
 *     BankAccountSurgeModel.scala:22-32 — restated in oracle.py (play_json_double_text) from the published pieces:
 *     java.lang.Double.toString's shortest digits (JDK >= 19; pinned here against Python's repr, the same shortest
 *     round-trip digits, on tens of thousands of random doubles plus subnormal / boundary cases), BigDecimal.valueOf +
 *     stripTrailingZeros + toPlainString / toString switch as play-json's JsNumber writer applies them; the product's
 *     GPU and host formatters are checked against THAT (tests/test_f64_text.py), no JVM-produced text exists here.
 *
 * Everything here is written as a literal, sequential, one-event-at-a-time
 * reading of the Scala code it cites — deliberately NOT the transformer-monoid
 * formulation the GPU kernels use, so that agreement between the two means
 * something.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdio.h>
#include <pthread.h>

#include "../include/surge_replay.h"

/* Option[State]: `present` is Some/None; `poisoned` models "handleEvent threw". */
typedef struct {
  int present;
  int poisoned;
  int32_t count, version;
  int64_t sum64;
  uint64_t balance_bits;
  int32_t min_arg, max_arg;
  uint32_t event_count;
} opt_state;

static void from_state64(const surge_state64* s, opt_state* o) {
  uint64_t bits;
  memcpy(&bits, &s->balance, 8);
  o->present = (s->flags & SURGE_STATE_PRESENT) != 0;
  o->poisoned = (s->flags & SURGE_STATE_POISONED) != 0;
  o->count = s->count;
  o->version = s->version;
  o->sum64 = s->sum64;
  o->balance_bits = bits;
  o->min_arg = s->min_arg;
  o->max_arg = s->max_arg;
  o->event_count = s->event_count;
}

static void to_state64(const opt_state* o, surge_state64* s) {
  memset(s, 0, sizeof(*s));
  if (o->present) {
    s->count = o->count;
    s->version = o->version;
    s->sum64 = o->sum64;
    memcpy(&s->balance, &o->balance_bits, 8);
    s->min_arg = o->min_arg;
    s->max_arg = o->max_arg;
    s->event_count = o->event_count;
    s->flags |= SURGE_STATE_PRESENT;
  }
  if (o->poisoned) s->flags |= SURGE_STATE_POISONED;
}

/* `agg.getOrElse(State(evt.aggregateId, 0, 0))` — TestBoundedContext.scala:78 */
static void materialise_default(const surge_replay_schema* sc, opt_state* o) {
  opt_state d;
  from_state64(&sc->default_state, &d);
  o->present = 1;
  o->count = d.count;
  o->version = d.version;
  o->sum64 = d.sum64;
  o->balance_bits = d.balance_bits;
  o->min_arg = d.min_arg;
  o->max_arg = d.max_arg;
  o->event_count = d.event_count;
}

/* The field updates of one case of handleEvent, applied to a Some(current).
 * JVM Int arithmetic wraps mod 2^32 (TestBoundedContext.scala:82,84): do the
 * adds in uint32 so C has no signed-overflow UB. */
static void update_fields(uint32_t d, const surge_event16* ev, opt_state* o) {
  switch (d & SURGE_D_COUNT_MASK) {
    case SURGE_D_COUNT_ADD: o->count = (int32_t)((uint32_t)o->count + (uint32_t)ev->p.i.arg); break;
    case SURGE_D_COUNT_SUB: o->count = (int32_t)((uint32_t)o->count - (uint32_t)ev->p.i.arg); break;
    case SURGE_D_COUNT_SET: o->count = ev->p.i.arg; break;
    default: break;
  }
  if (d & SURGE_D_VERSION_SET) o->version = ev->seq;
  switch (d & SURGE_D_SUM_MASK) {
    case SURGE_D_SUM_ADD: o->sum64 = (int64_t)((uint64_t)o->sum64 + (uint64_t)(int64_t)ev->p.i.arg); break;
    case SURGE_D_SUM_SUB: o->sum64 = (int64_t)((uint64_t)o->sum64 - (uint64_t)(int64_t)ev->p.i.arg); break;
    default: break;
  }
  if (d & SURGE_D_BALANCE_SET) o->balance_bits = ev->p.raw;
  if ((d & SURGE_D_MIN_ARG) && ev->p.i.arg < o->min_arg) o->min_arg = ev->p.i.arg;
  if ((d & SURGE_D_MAX_ARG) && ev->p.i.arg > o->max_arg) o->max_arg = ev->p.i.arg;
  if (d & SURGE_D_EVCOUNT_INC) o->event_count += 1u;
}

/*
 * handleEvent(aggregate: Option[Agg], event: Evt): Option[Agg]
 * — CommandModels.scala:14, as declared by the schema's descriptor for the
 * event's type.  Returns nonzero when the event "throws".
 *
 *   MATERIALIZE  TestBoundedContext.scala:77-89
 *   REQUIRE      BankAccountCommandModel.scala:84
 *   CREATE       BankAccountCommandModel.scala:83
 *   DELETE       Option[Agg] = None (tombstone, SurgeModel.scala:62)
 *   throw        TestBoundedContext.scala:86 ; caller keeps the old state
 *                (PersistentActor.scala:260-263)
 */
static int handle_event(const surge_replay_schema* sc, opt_state* agg, const surge_event16* ev) {
  uint32_t d;
  if (ev->type < 0 || (uint32_t)ev->type >= sc->n_types) return 1; /* no such case: MatchError */
  d = sc->desc[ev->type];
  if (d & SURGE_D_POISON) return 1;
  switch (d & SURGE_CLS_MASK) {
    case SURGE_CLS_DELETE:
      agg->present = 0;
      return 0;
    case SURGE_CLS_REQUIRE:
      if (!agg->present) return 0; /* aggregate.map(...) on None */
      update_fields(d, ev, agg);
      return 0;
    case SURGE_CLS_CREATE:
      materialise_default(sc, agg); /* Some(<built only from the event>) */
      update_fields(d, ev, agg);
      return 0;
    default: /* SURGE_CLS_MATERIALIZE */
      if (!agg->present) materialise_default(sc, agg);
      update_fields(d, ev, agg);
      return 0;
  }
}

/* One step, exposed for the known-answer tests. */
int32_t oracle_handle_event(const surge_replay_schema* sc, const surge_state64* in,
                            const surge_event16* ev, surge_state64* out) {
  opt_state o;
  from_state64(in, &o);
  if (!o.poisoned && handle_event(sc, &o, ev)) o.poisoned = 1;
  to_state64(&o, out);
  return 0;
}

/*
 * events.foldLeft(state)((stateAccum, evt) => handleEvent(stateAccum, evt))
 * — CommandModels.scala:26 — for aggregates [a0, a1).  Replay semantics for a
 * throwing event (the reference defines none for replay; PersistentActor.scala:
 * 260-263 keeps the pre-event state): stop at the first throwing event, keep the
 * state before it, flag the aggregate POISONED.
 */
static void fold_range(const surge_replay_schema* sc, const int64_t* seg_off, int64_t a0, int64_t a1,
                       const surge_event16* events, const surge_state64* init, surge_state64* out) {
  int64_t a, e;
  for (a = a0; a < a1; ++a) {
    opt_state acc;
    if (init) {
      from_state64(&init[a], &acc);
    } else {
      memset(&acc, 0, sizeof(acc)); /* None */
    }
    for (e = seg_off[a]; e < seg_off[a + 1] && !acc.poisoned; ++e) {
      if (handle_event(sc, &acc, &events[e])) acc.poisoned = 1;
    }
    to_state64(&acc, &out[a]);
  }
}

int32_t oracle_fold_csr(const surge_replay_schema* sc, const int64_t* seg_off, int64_t n_agg,
                        const void* events, const void* init_state, void* out_states) {
  if (!sc || !seg_off || n_agg < 0 || !out_states) return -1;
  fold_range(sc, seg_off, 0, n_agg, (const surge_event16*)events, (const surge_state64*)init_state,
             (surge_state64*)out_states);
  return 0;
}

/* ---------------------------------------------------------------------------
 * ABI v2 slot schemas (include/surge_replay.h): the same fold, handleEvent restated per event type as one
 * operation per typed slot.  Literal and sequential: one event at a time, doubles added in event order
 * (built with -ffp-contract=off), exactly what events.foldLeft(state)(handleEvent) does on the JVM.
 * ------------------------------------------------------------------------- */
static uint64_t slot_operand(const surge_slot_def* sd, const surge_event16* ev) {
  if (sd->type == SURGE_SLOT_F64) {
    double d;
    uint64_t bits;
    if (sd->source == SURGE_SRC_PAYLOAD) return ev->p.raw;
    d = sd->source == SURGE_SRC_ONE ? 1.0 : (double)(sd->source == SURGE_SRC_SEQ ? ev->seq : ev->p.i.arg);
    memcpy(&bits, &d, 8);
    return bits;
  } else {
    int64_t v = sd->source == SURGE_SRC_PAYLOAD ? (int64_t)ev->p.raw
              : sd->source == SURGE_SRC_ONE ? 1
              : (int64_t)(sd->source == SURGE_SRC_SEQ ? ev->seq : ev->p.i.arg);
    return sd->type == SURGE_SLOT_I32 ? (uint64_t)(uint32_t)v : (uint64_t)v;
  }
}

static uint64_t slot_apply(const surge_slot_def* sd, uint32_t op, uint64_t cur, uint64_t x) {
  if (op == SURGE_OP_KEEP) return cur;
  if (op == SURGE_OP_SET) return x;
  if (sd->type == SURGE_SLOT_F64) {
    double a, b, r;
    uint64_t bits;
    memcpy(&a, &cur, 8);
    memcpy(&b, &x, 8);
    switch (op) {
      case SURGE_OP_ADD: r = a + b; break;
      case SURGE_OP_SUB: r = a - b; break;
      /* java.lang.Math.min(a, b) / Math.max(a, b) as the JDK library source states them (what a Scala model's
       * math.min / math.max calls): `if (a != a) return a;` then the signed-zero case, then `(a <= b) ? a : b` /
       * `(a >= b) ? a : b` — so a NaN operand b is returned as is, and -0.0 orders below +0.0. */
      case SURGE_OP_MIN:
        if (a != a) return cur;
        if (a == 0.0 && b == 0.0 && x == 0x8000000000000000ull) return x;
        return (a <= b) ? cur : x;
      default: /* SURGE_OP_MAX */
        if (a != a) return cur;
        if (a == 0.0 && b == 0.0 && cur == 0x8000000000000000ull) return x;
        return (a >= b) ? cur : x;
    }
    memcpy(&bits, &r, 8);
    return bits;
  }
  if (sd->type == SURGE_SLOT_I64) {
    switch (op) {
      case SURGE_OP_ADD: return cur + x;
      case SURGE_OP_SUB: return cur - x;
      case SURGE_OP_MIN: return (int64_t)x < (int64_t)cur ? x : cur;
      default: return (int64_t)x > (int64_t)cur ? x : cur;
    }
  }
  {
    uint32_t a = (uint32_t)cur, b = (uint32_t)x;
    switch (op) {
      case SURGE_OP_ADD: return (uint64_t)(uint32_t)(a + b);
      case SURGE_OP_SUB: return (uint64_t)(uint32_t)(a - b);
      case SURGE_OP_MIN: return (int32_t)b < (int32_t)a ? (uint64_t)b : (uint64_t)a;
      default: return (int32_t)b > (int32_t)a ? (uint64_t)b : (uint64_t)a;
    }
  }
}

typedef struct { uint64_t s[SURGE_MAX_SLOTS]; uint32_t evc; int present, poisoned; } slot_state;

static void slots_from64(const uint8_t* in, slot_state* o) {
  uint32_t fl;
  int i;
  for (i = 0; i < SURGE_MAX_SLOTS; ++i) memcpy(&o->s[i], in + SURGE_SLOT_OFFSET(i), 8);
  memcpy(&o->evc, in + 32, 4);
  memcpy(&fl, in + 36, 4);
  o->present = (fl & SURGE_STATE_PRESENT) != 0;
  o->poisoned = (fl & SURGE_STATE_POISONED) != 0;
}

static void slots_to64(const slot_state* o, uint8_t* out) {
  uint32_t fl = (o->present ? SURGE_STATE_PRESENT : 0u) | (o->poisoned ? SURGE_STATE_POISONED : 0u);
  int i;
  memset(out, 0, 64); /* None is canonically all-zero */
  if (o->present) {
    for (i = 0; i < SURGE_MAX_SLOTS; ++i) memcpy(out + SURGE_SLOT_OFFSET(i), &o->s[i], 8);
    memcpy(out + 32, &o->evc, 4);
  }
  memcpy(out + 36, &fl, 4);
}

/* returns nonzero when the event "throws" */
static int handle_event_v2(const surge_replay_schema_v2* sc, slot_state* agg, const surge_event16* ev) {
  uint32_t cls, ops, c, i;
  if (ev->type < 0 || (uint32_t)ev->type >= sc->n_types) return 1; /* MatchError */
  cls = sc->cls[ev->type];
  ops = sc->ops[ev->type];
  if (cls & SURGE_D_POISON) return 1;
  c = cls & SURGE_CLS_MASK;
  if (c == SURGE_CLS_DELETE) { agg->present = 0; return 0; }
  if (c == SURGE_CLS_REQUIRE && !agg->present) return 0;
  if (c == SURGE_CLS_CREATE || !agg->present) { /* Some(<built from the event>) / getOrElse(default) */
    for (i = 0; i < sc->n_slots; ++i)
      agg->s[i] = sc->slot[i].type == SURGE_SLOT_I32 ? (uint64_t)(uint32_t)sc->slot[i].default_bits : sc->slot[i].default_bits;
    for (i = sc->n_slots; i < SURGE_MAX_SLOTS; ++i) agg->s[i] = 0;
    agg->evc = 0;
    agg->present = 1;
  }
  if (sc->flags & SURGE_V2_COUNT_EVENTS) agg->evc += 1u;
  for (i = 0; i < sc->n_slots; ++i)
    agg->s[i] = slot_apply(&sc->slot[i], (ops >> (4 * i)) & 15u, agg->s[i], slot_operand(&sc->slot[i], ev));
  return 0;
}

int32_t oracle_fold_csr_v2(const surge_replay_schema_v2* sc, const int64_t* seg_off, int64_t n_agg, const void* events,
                           const void* init_state, void* out_states) {
  const surge_event16* ev = (const surge_event16*)events;
  const uint8_t* init = (const uint8_t*)init_state;
  uint8_t* out = (uint8_t*)out_states;
  int64_t a, e;
  if (!sc || !seg_off || n_agg < 0 || !out_states) return -1;
  for (a = 0; a < n_agg; ++a) {
    slot_state acc;
    if (init) slots_from64(init + a * 64, &acc);
    else memset(&acc, 0, sizeof(acc));
    for (e = seg_off[a]; e < seg_off[a + 1] && !acc.poisoned; ++e)
      if (handle_event_v2(sc, &acc, &ev[e])) acc.poisoned = 1;
    slots_to64(&acc, out + a * 64);
  }
  return 0;
}

/* Same fold with aggregates split over host threads (baseline B2). */
typedef struct {
  const surge_replay_schema* sc;
  const int64_t* seg_off;
  int64_t a0, a1;
  const surge_event16* events;
  const surge_state64* init;
  surge_state64* out;
  int32_t reps;
} fold_job;

static void* fold_thread(void* p) {
  fold_job* j = (fold_job*)p;
  int32_t r;
  for (r = 0; r < j->reps; ++r) fold_range(j->sc, j->seg_off, j->a0, j->a1, j->events, j->init, j->out);
  return NULL;
}

/* `reps` full folds per thread over its range (same result every time): lets a benchmark time sustained all-core
 * folding without paying a pthread_create per pass (bench.py's cpu_baseline). */
int32_t oracle_fold_csr_mt_reps(const surge_replay_schema* sc, const int64_t* seg_off, int64_t n_agg,
                                const void* events, const void* init_state, void* out_states,
                                int32_t n_threads, int32_t reps);

int32_t oracle_fold_csr_mt(const surge_replay_schema* sc, const int64_t* seg_off, int64_t n_agg,
                           const void* events, const void* init_state, void* out_states,
                           int32_t n_threads) {
  return oracle_fold_csr_mt_reps(sc, seg_off, n_agg, events, init_state, out_states, n_threads, 1);
}

int32_t oracle_fold_csr_mt_reps(const surge_replay_schema* sc, const int64_t* seg_off, int64_t n_agg,
                                const void* events, const void* init_state, void* out_states,
                                int32_t n_threads, int32_t reps) {
  enum { MAXT = 256 };
  pthread_t th[MAXT];
  fold_job jobs[MAXT];
  int t, started = 0;
  int64_t total, target, a;
  if (!sc || !seg_off || n_agg < 0 || !out_states) return -1;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > MAXT) n_threads = MAXT;
  total = seg_off[n_agg];
  a = 0;
  for (t = 0; t < n_threads; ++t) {
    int64_t a_end;
    /* split by events, not aggregates, so Zipf logs balance */
    target = (t + 1 == n_threads) ? total : (total / n_threads) * (t + 1);
    a_end = a;
    if (t + 1 == n_threads) {
      a_end = n_agg;
    } else {
      while (a_end < n_agg && seg_off[a_end] < target) ++a_end;
    }
    jobs[t].sc = sc;
    jobs[t].seg_off = seg_off;
    jobs[t].a0 = a;
    jobs[t].a1 = a_end;
    jobs[t].events = (const surge_event16*)events;
    jobs[t].init = (const surge_state64*)init_state;
    jobs[t].out = (surge_state64*)out_states;
    jobs[t].reps = reps < 1 ? 1 : reps;
    a = a_end;
  }
  for (t = 0; t < n_threads; ++t) {
    if (pthread_create(&th[t], NULL, fold_thread, &jobs[t]) != 0) break;
    ++started;
  }
  for (t = started; t < n_threads; ++t) fold_thread(&jobs[t]); /* degraded: run inline */
  for (t = 0; t < started; ++t) pthread_join(th[t], NULL);
  return 0;
}

/* ---------------------------------------------------------------------------
 * scala.util.hashing.MurmurHash3.stringHash(str)  [scala-library 2.13.8, not
 * vendored under /root/reference; call site KafkaPartitioner.scala:8].
 * Published algorithm: seed 0xf7ca7fd2; UTF-16 code units consumed in pairs
 * data = (c[i] << 16) + c[i+1]; odd tail through mixLast; finalizeHash(h, len).
 * PARITY UNPINNED: no reference test fixes a value of this function.
 * ------------------------------------------------------------------------- */
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

static uint32_t mm3_mix_last(uint32_t hash, uint32_t data) {
  uint32_t k = data;
  k *= 0xcc9e2d51u;
  k = rotl32(k, 15);
  k *= 0x1b873593u;
  return hash ^ k;
}

static uint32_t mm3_mix(uint32_t hash, uint32_t data) {
  uint32_t h = mm3_mix_last(hash, data);
  h = rotl32(h, 13);
  return h * 5u + 0xe6546b64u;
}

static uint32_t mm3_finalize(uint32_t hash, uint32_t length) {
  uint32_t h = hash ^ length;
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

int32_t oracle_murmur3_string_hash(const uint16_t* s, int64_t len) {
  uint32_t h = 0xf7ca7fd2u;
  int64_t i = 0;
  while (i + 1 < len) {
    uint32_t data = ((uint32_t)s[i] << 16) + (uint32_t)s[i + 1];
    h = mm3_mix(h, data);
    i += 2;
  }
  if (i < len) h = mm3_mix_last(h, (uint32_t)s[i]);
  return (int32_t)mm3_finalize(h, (uint32_t)len);
}

/* KafkaPartitionProvider.partitionForKey(partitionByString, numberOfPartitions) =
 * math.abs(MurmurHash3.stringHash(partitionByString) % numberOfPartitions) — KafkaPartitioner.scala:8:
 * the WHOLE string.  JVM `%` truncates toward zero, like C99. */
int32_t oracle_partition_for_key(const uint16_t* s, int64_t len, int32_t n_partitions) {
  const int32_t h = oracle_murmur3_string_hash(s, len);
  const int32_t r = h % n_partitions;
  return r < 0 ? -r : r;
}

/* PartitionStringUpToColon.partitionBy = str.takeWhile(_ != ':') — KafkaPartitioner.scala:38-42.
 * Returns the length of the prefix. */
int64_t oracle_partition_by_up_to_colon(const uint16_t* s, int64_t len) {
  int64_t n = 0;
  while (n < len && s[n] != (uint16_t)':') ++n;
  return n;
}

int32_t oracle_partition_hash_batch(const uint16_t* utf16, const int64_t* str_off, int64_t n,
                                    int32_t n_partitions, int32_t up_to_colon, int32_t* part_out) {
  int64_t i;
  if (n_partitions <= 0) return -1;
  for (i = 0; i < n; ++i) {
    const uint16_t* s = utf16 + str_off[i];
    int64_t len = str_off[i + 1] - str_off[i];
    if (up_to_colon) len = oracle_partition_by_up_to_colon(s, len);
    part_out[i] = oracle_partition_for_key(s, len, n_partitions);
  }
  return 0;
}

/* Standard MurmurHash3_x86_32 over bytes (Appleby's published algorithm), built from the SAME mix /
 * mixLast / finalize primitives as oracle_murmur3_string_hash above, so that the published verification
 * values of MurmurHash3_x86_32 pin those primitives (tests/test_oracle_kat.py). */
uint32_t oracle_murmur3_x86_32(const uint8_t* data, int64_t len, uint32_t seed) {
  uint32_t h = seed, k = 0;
  int64_t i = 0;
  for (; i + 4 <= len; i += 4) {
    k = (uint32_t)data[i] | ((uint32_t)data[i + 1] << 8) | ((uint32_t)data[i + 2] << 16) | ((uint32_t)data[i + 3] << 24);
    h = mm3_mix(h, k);
  }
  k = 0;
  switch (len & 3) {
    case 3: k ^= (uint32_t)data[i + 2] << 16; /* fallthrough */
    case 2: k ^= (uint32_t)data[i + 1] << 8;  /* fallthrough */
    case 1: k ^= (uint32_t)data[i];
            h = mm3_mix_last(h, k);
  }
  return mm3_finalize(h, (uint32_t)len);
}

/* ---------------------------------------------------------------------------
 * Json.toJson(agg).toString().getBytes() for State(aggregateId, count, version)
 * — TestBoundedContext.scala:15-16,127-129.  play-json's Json.format macro
 * emits fields in declaration order, compact: {"aggregateId":"…","count":4,"version":4}
 * (the shape asserted by AggregateStateStoreKafkaStreamsSpec.scala:64-85).
 * String escaping follows Jackson's default (\" \\ \b \f \n \r \t, other
 * controls as \u00XX, non-ASCII passed through as UTF-8).
 * Returns the number of bytes written (excluding NUL), or -1 if cap is too small.
 * ------------------------------------------------------------------------- */
static int64_t put(char* out, int64_t cap, int64_t pos, const char* s, int64_t n) {
  if (pos < 0 || pos + n >= cap) return -1;
  memcpy(out + pos, s, (size_t)n);
  return pos + n;
}

static int64_t put_json_string(char* out, int64_t cap, int64_t pos, const char* utf8) {
  const unsigned char* p = (const unsigned char*)utf8;
  char buf[8];
  pos = put(out, cap, pos, "\"", 1);
  for (; *p && pos >= 0; ++p) {
    switch (*p) {
      case '"': pos = put(out, cap, pos, "\\\"", 2); break;
      case '\\': pos = put(out, cap, pos, "\\\\", 2); break;
      case '\b': pos = put(out, cap, pos, "\\b", 2); break;
      case '\f': pos = put(out, cap, pos, "\\f", 2); break;
      case '\n': pos = put(out, cap, pos, "\\n", 2); break;
      case '\r': pos = put(out, cap, pos, "\\r", 2); break;
      case '\t': pos = put(out, cap, pos, "\\t", 2); break;
      default:
        if (*p < 0x20) {
          snprintf(buf, sizeof(buf), "\\u%04X", (unsigned)*p);
          pos = put(out, cap, pos, buf, 6);
        } else {
          pos = put(out, cap, pos, (const char*)p, 1);
        }
    }
  }
  if (pos >= 0) pos = put(out, cap, pos, "\"", 1);
  return pos;
}

int64_t oracle_counter_state_json(const char* aggregate_id_utf8, int32_t count, int32_t version,
                                  char* out, int64_t cap) {
  char num[32];
  int64_t pos = 0;
  int n;
  pos = put(out, cap, pos, "{\"aggregateId\":", 15);
  if (pos >= 0) pos = put_json_string(out, cap, pos, aggregate_id_utf8);
  if (pos >= 0) pos = put(out, cap, pos, ",\"count\":", 9);
  n = snprintf(num, sizeof(num), "%d", count);
  if (pos >= 0) pos = put(out, cap, pos, num, n);
  if (pos >= 0) pos = put(out, cap, pos, ",\"version\":", 11);
  n = snprintf(num, sizeof(num), "%d", version);
  if (pos >= 0) pos = put(out, cap, pos, num, n);
  if (pos >= 0) pos = put(out, cap, pos, "}", 1);
  if (pos >= 0) out[pos] = 0;
  return pos;
}
