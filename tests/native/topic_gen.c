/* TEST / BENCH INFRASTRUCTURE (not part of the product): the events topic's record keys and values for the Counter
 * fixture, in bulk — what `CounterEventFormat.write_event` (examples/fixture_models.py, restating
 * TestBoundedContext.scala:42-49,122-124) writes per event, for arrays of (aggregate number, type, argument, sequence
 * number).  bench.py --workload e2e and tests/test_bench_rehearsal.py synthesise topics of 10^8 records with it; a
 * sample of its output is compared with write_event byte for byte wherever it is used.
 *   key   = "acct-%08d:<seq>"
 *   value = {"aggregateId":"acct-%08d","incrementBy":<arg>,"sequenceNumber":<seq>,"_type":"countIncremented"}   type 0
 *           {"aggregateId":"acct-%08d","decrementBy":<arg>,"sequenceNumber":<seq>,"_type":"countDecremented"}   type 1
 *           {"aggregateId":"acct-%08d","sequenceNumber":<seq>,"_type":"no-op"}                                   type 2
 * gcc -O2 -shared -fPIC tests/native/topic_gen.c -o tests/native/libtopic_gen.so */
#include <stdint.h>
#include <string.h>

static uint8_t* put(uint8_t* p, const char* s) {
  const size_t n = strlen(s);
  memcpy(p, s, n);
  return p + n;
}
static uint8_t* put_u(uint8_t* p, uint64_t v) {
  char tmp[24];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *p++ = (uint8_t)tmp[--n];
  return p;
}
static uint8_t* put_i(uint8_t* p, int64_t v) {
  if (v < 0) { *p++ = '-'; return put_u(p, (uint64_t)(-v)); }
  return put_u(p, (uint64_t)v);
}
static uint8_t* put_id(uint8_t* p, int64_t agg) {
  p = put(p, "acct-");
  for (int k = 7; k >= 0; --k) { p[k] = (uint8_t)('0' + agg % 10); agg /= 10; }
  return p + 8;
}

static int digits_u(uint64_t v) { int n = 1; while (v >= 10) { v /= 10; ++n; } return n; }
static int digits_i(int64_t v) { return v < 0 ? 1 + digits_u((uint64_t)(-v)) : digits_u((uint64_t)v); }

/* keys / vals need 40 / 128 bytes per record at most; key_off / val_off have n + 1 entries.  Returns n.  Two passes:
 * the lengths (then their running sums), then the text — the second one side by side (OpenMP, when compiled with it). */
int64_t surge_test_counter_records(int64_t n, const int64_t* agg, const int32_t* type, const int32_t* arg, const int32_t* seq, uint8_t* keys,
                                   int64_t* key_off, uint8_t* vals, int64_t* val_off) {
  key_off[0] = 0;
  val_off[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int ds = digits_i(seq[i]);
    key_off[i + 1] = key_off[i] + 13 + 1 + ds;
    /* {"aggregateId":"acct-%08d", = 16 + 13 + 2;  "xxcrementBy":<arg>, = 14 + digits + 1;  "sequenceNumber":<seq> = 17 + digits;  ,"_type":" = 10;  name;  "} = 2 */
    int64_t vl = 31 + 17 + ds + 10 + 2;
    if (type[i] == 0 || type[i] == 1) vl += 15 + digits_i(arg[i]) + 16;
    else vl += 5;
    val_off[i + 1] = val_off[i] + vl;
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    uint8_t* k = keys + key_off[i];
    uint8_t* v = vals + val_off[i];
    k = put_id(k, agg[i]);
    *k++ = ':';
    k = put_i(k, seq[i]);
    v = put(v, "{\"aggregateId\":\"");
    v = put_id(v, agg[i]);
    v = put(v, "\",");
    if (type[i] == 0) { v = put(v, "\"incrementBy\":"); v = put_i(v, arg[i]); *v++ = ','; }
    if (type[i] == 1) { v = put(v, "\"decrementBy\":"); v = put_i(v, arg[i]); *v++ = ','; }
    v = put(v, "\"sequenceNumber\":");
    v = put_i(v, seq[i]);
    v = put(v, ",\"_type\":\"");
    v = put(v, type[i] == 0 ? "countIncremented" : type[i] == 1 ? "countDecremented" : "no-op");
    v = put(v, "\"}");
    if (k != keys + key_off[i + 1] || v != vals + val_off[i + 1]) __builtin_trap(); /* the length pass and the text disagree */
  }
  return n;
}

/* ---- BankAccount (surge-docs sample) -----------------------------------------------------------------------------------
 * What `BankAccountEventFormat.write_event` (examples/fixture_models.py, restating BankAccountSurgeModel.scala:30-32)
 * writes: the record key is the account's UUID alone (no ':' and no sequence number), the balances are JSON numbers as
 * play-json writes a Double.
 *   account number of aggregate a:  hhhhhhhh-hhhh-4hhh-[89ab]hhh-<a as 12 hex digits>   (h: a hash of a; version 4, variant 1)
 *   j = 1:  {"accountNumber":"<uuid>","accountOwner":"Owner <a % 1000>","securityCode":"<a % 10000, 4 digits>","balance":<amount>,"_type":"docs.command.BankAccountCreated"}
 *   j > 1:  {"accountNumber":"<uuid>","newBalance":<amount>,"_type":"docs.command.BankAccountUpdated"}
 * amount = cents / 100 (0 < cents < 10^9: Double.toString stays in plain notation), written with its shortest digits:
 * 1000, 1234.5, 12.34.  keys need 36 bytes per record, vals 192. */
static uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static uint8_t* put_hex(uint8_t* p, uint64_t v, int digits) {
  for (int k = digits - 1; k >= 0; --k) { p[k] = (uint8_t)"0123456789abcdef"[v & 15]; v >>= 4; }
  return p + digits;
}
static uint8_t* put_uuid(uint8_t* p, int64_t agg) {
  const uint64_t h = mix64((uint64_t)agg), g = mix64(h);
  p = put_hex(p, h >> 32, 8); *p++ = '-';
  p = put_hex(p, (h >> 16) & 0xffff, 4); *p++ = '-';
  p = put_hex(p, 0x4000 | (h & 0x0fff), 4); *p++ = '-';
  p = put_hex(p, 0x8000 | (g & 0x3fff), 4); *p++ = '-';
  return put_hex(p, (uint64_t)agg, 12);
}
static int amount_len(int64_t cents) {
  const int fr = (int)(cents % 100);
  return digits_u((uint64_t)(cents / 100)) + (fr == 0 ? 0 : fr % 10 == 0 ? 2 : 3);
}
static uint8_t* put_amount(uint8_t* p, int64_t cents) {
  const int fr = (int)(cents % 100);
  p = put_u(p, (uint64_t)(cents / 100));
  if (fr == 0) return p;
  *p++ = '.';
  *p++ = (uint8_t)('0' + fr / 10);
  if (fr % 10) *p++ = (uint8_t)('0' + fr % 10);
  return p;
}
int64_t surge_test_bank_records(int64_t n, const int64_t* agg, const int32_t* seq, const int64_t* cents, uint8_t* keys, int64_t* key_off, uint8_t* vals,
                                int64_t* val_off) {
  key_off[0] = 0;
  val_off[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    key_off[i + 1] = key_off[i] + 36;
    /* {"accountNumber":" = 18; uuid 36; ", = 2 */
    int64_t vl = 18 + 36 + 2;
    if (seq[i] == 1) /* "accountOwner":"Owner  = 22; n; ","securityCode":" = 18; 4; ","balance": = 12; amount; ,"_type":" = 10; name 31; "} = 2 */
      vl += 22 + digits_u((uint64_t)(agg[i] % 1000)) + 18 + 4 + 12 + amount_len(cents[i]) + 10 + 31 + 2;
    else /* "newBalance": = 13; amount; ,"_type":" = 10; name 31; "} = 2 */
      vl += 13 + amount_len(cents[i]) + 10 + 31 + 2;
    val_off[i + 1] = val_off[i] + vl;
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    uint8_t* k = keys + key_off[i];
    uint8_t* v = vals + val_off[i];
    k = put_uuid(k, agg[i]);
    v = put(v, "{\"accountNumber\":\"");
    v = put_uuid(v, agg[i]);
    v = put(v, "\",");
    if (seq[i] == 1) {
      v = put(v, "\"accountOwner\":\"Owner ");
      v = put_u(v, (uint64_t)(agg[i] % 1000));
      v = put(v, "\",\"securityCode\":\"");
      for (int d = 3, c = (int)(agg[i] % 10000); d >= 0; --d) { v[d] = (uint8_t)('0' + c % 10); c /= 10; }
      v += 4;
      v = put(v, "\",\"balance\":");
      v = put_amount(v, cents[i]);
      v = put(v, ",\"_type\":\"docs.command.BankAccountCreated\"}");
    } else {
      v = put(v, "\"newBalance\":");
      v = put_amount(v, cents[i]);
      v = put(v, ",\"_type\":\"docs.command.BankAccountUpdated\"}");
    }
    if (k != keys + key_off[i + 1] || v != vals + val_off[i + 1]) __builtin_trap(); /* the length pass and the text disagree */
  }
  return n;
}
