// event_decode.cpp — events-topic record VALUES as the reference's plugins write them (JSON text) -> the 16-byte fixed
// event the fold takes.  The mirror image of the state encoders (fixed 64-byte state -> text): template driven, no
// per-record host-language callback.
//
// What is decoded: the reference's event writers are `Json.toJson(evt).toString().getBytes()` over play-json `Json.format`
// case-class formats (modules/command-engine/scaladsl/src/test/scala/surge/scaladsl/TestBoundedContext.scala:42-49,
// 122-124; modules/surge-docs/src/test/scala/docs/command/BankAccountSurgeModel.scala:30-32): one flat JSON object per
// event — string / integer / decimal fields in declaration order, and for a sealed family a discriminator field
// ("_type") naming the case class.  The decoder does not depend on field order or on fields it is not told about: it
// scans the top-level object once, remembers the discriminator string and every numeric field, then builds the event from
// the template entry the discriminator selects.  Numbers: an I32 payload / the sequence number must be a JSON integer
// that fits an Int (anything else is reported, not truncated); an F64 payload is converted by
// surge_parse_f64_json (f64_parse.h), correctly rounded like the JVM's BigDecimal.doubleValue that play-json reads Doubles with.
//
// play-json itself is not under /root/reference (com.typesafe.play:play-json 2.9.2): its exact text — where it puts the
// discriminator, how it spells a Double — is parity-unpinned; being order- and spelling-agnostic is what
// keeps the decoder correct either way.
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/surge_ingest.h"

namespace {

struct Field {
  const uint8_t* key;
  int key_len;
  const uint8_t* num;  // numeric literal (nullptr: not a number)
  int num_len;
  const uint8_t* str;  // string contents without the quotes (nullptr: not a string)
  int str_len;
  bool str_escaped;
};

constexpr int kMaxFields = 24;

struct Scanner {
  const uint8_t* p;
  const uint8_t* end;
  void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
  // string starting at the opening quote; returns false on a malformed / unterminated string
  bool string(const uint8_t** s, int* len, bool* escaped) {
    if (p >= end || *p != '"') return false;
    ++p;
    *s = p;
    *escaped = false;
    while (p < end && *p != '"') {
      if (*p == '\\') {
        *escaped = true;
        ++p;
        if (p >= end) return false;
      }
      ++p;
    }
    if (p >= end) return false;
    *len = (int)(p - *s);
    ++p;
    return true;
  }
  bool number(const uint8_t** s, int* len) {
    *s = p;
    if (p < end && (*p == '-' || *p == '+')) ++p;
    bool digits = false;
    while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) {
      digits = digits || (*p >= '0' && *p <= '9');
      ++p;
    }
    *len = (int)(p - *s);
    return digits;
  }
  // skips one value of any kind (nested containers by depth counting, strings honoured)
  bool skip_value() {
    ws();
    if (p >= end) return false;
    if (*p == '"') {
      const uint8_t* s; int l; bool e;
      return string(&s, &l, &e);
    }
    if (*p == '{' || *p == '[') {
      int depth = 0;
      while (p < end) {
        if (*p == '"') {
          const uint8_t* s; int l; bool e;
          if (!string(&s, &l, &e)) return false;
          continue;
        }
        if (*p == '{' || *p == '[') ++depth;
        if (*p == '}' || *p == ']') {
          --depth;
          if (depth == 0) { ++p; return true; }
        }
        ++p;
      }
      return false;
    }
    if (*p == 't' || *p == 'f' || *p == 'n') {
      while (p < end && *p >= 'a' && *p <= 'z') ++p;
      return true;
    }
    const uint8_t* s; int l;
    return number(&s, &l);
  }
};

bool name_is(const char* want, const uint8_t* got, int got_len) {
  const size_t n = std::strlen(want);
  return n == (size_t)got_len && std::memcmp(want, got, n) == 0;
}

// 0 ok, 1 not an integer literal, 2 out of int32 range
int parse_i32(const uint8_t* s, int len, int32_t* out) {
  if (len <= 0 || len > 11) return len > 11 ? 2 : 1;
  int i = 0;
  bool neg = false;
  if (s[0] == '-') { neg = true; i = 1; }
  if (i >= len) return 1;
  int64_t v = 0;
  for (; i < len; ++i) {
    if (s[i] < '0' || s[i] > '9') return 1;
    v = v * 10 + (s[i] - '0');
  }
  if (neg) v = -v;
  if (v < INT32_MIN || v > INT32_MAX) return 2;
  *out = (int32_t)v;
  return 0;
}

bool parse_f64(const uint8_t* s, int len, double* out) {
  // the library's own correctly rounded parser (Eisel-Lemire, strtod where that cannot decide): the code the device
  // decoder runs, so host and device agree on every value by construction — and both with BigDecimal.doubleValue
  uint64_t bits = 0;
  if (surge_parse_f64_json(s, len, &bits) < 0) return false;
  std::memcpy(out, &bits, 8);
  return true;
}

thread_local std::string t_err;

}  // namespace

extern "C" {

const char* surge_event_json_last_error(void) { return t_err.c_str(); }

int32_t surge_event_json_validate(const surge_event_json_template* t) {
  auto bad = [](const char* m) { t_err = m; return -1; };
  if (!t) return bad("template is NULL");
  if (t->n_types < 1 || t->n_types > SURGE_EVJ_MAX_TYPES) return bad("template.n_types out of range");
  auto terminated = [](const char* s) { return std::memchr(s, 0, SURGE_EVJ_NAME) != nullptr; };
  if (!terminated(t->discriminator)) return bad("discriminator is not NUL-terminated");
  if (t->discriminator[0] == 0 && t->n_types != 1) return bad("a template without a discriminator has exactly one type");
  for (uint32_t i = 0; i < t->n_types; ++i) {
    const surge_event_json_type& e = t->types[i];
    if (!terminated(e.name) || !terminated(e.seq_field) || !terminated(e.arg_field)) return bad("a name is not NUL-terminated");
    if (e.arg_kind > SURGE_EVJ_ARG_F64) return bad("unknown arg_kind");
    if ((e.arg_kind == SURGE_EVJ_ARG_NONE) != (e.arg_field[0] == 0)) return bad("arg_field and arg_kind disagree");
  }
  return 0;
}

// One record value -> one 16-byte event.  0 = OK; SURGE_E_CORRUPT = not the JSON the template describes (message in
// surge_event_json_last_error).
int32_t surge_event_json_decode(const surge_event_json_template* t, const uint8_t* value, int64_t len, void* event16_out) {
  auto corrupt = [](const std::string& m) { t_err = m; return (int32_t)SURGE_E_CORRUPT; };
  if (!t || !event16_out || (!value && len > 0) || len < 0) { t_err = "bad argument"; return -1; }
  Scanner sc{value, value + len};
  sc.ws();
  if (sc.p >= sc.end || *sc.p != '{') return corrupt("event value is not a JSON object");
  ++sc.p;
  Field f[kMaxFields];
  int nf = 0;
  sc.ws();
  if (sc.p < sc.end && *sc.p == '}') {
    ++sc.p;
  } else {
    for (;;) {
      sc.ws();
      Field cur{};
      bool esc = false;
      if (!sc.string(&cur.key, &cur.key_len, &esc)) return corrupt("malformed field name");
      sc.ws();
      if (sc.p >= sc.end || *sc.p != ':') return corrupt("missing ':' after a field name");
      ++sc.p;
      sc.ws();
      if (sc.p >= sc.end) return corrupt("truncated object");
      if (*sc.p == '"') {
        if (!sc.string(&cur.str, &cur.str_len, &cur.str_escaped)) return corrupt("unterminated string");
      } else if (*sc.p == '-' || (*sc.p >= '0' && *sc.p <= '9')) {
        if (!sc.number(&cur.num, &cur.num_len)) return corrupt("malformed number");
      } else if (!sc.skip_value()) {
        return corrupt("malformed value");
      }
      if (!esc && nf < kMaxFields) f[nf++] = cur;  // more fields than that: the later ones are ignored
      sc.ws();
      if (sc.p < sc.end && *sc.p == ',') { ++sc.p; continue; }
      if (sc.p < sc.end && *sc.p == '}') { ++sc.p; break; }
      return corrupt("expected ',' or '}'");
    }
  }
  sc.ws();
  if (sc.p != sc.end) return corrupt("trailing bytes after the event object");

  const surge_event_json_type* ty = nullptr;
  if (t->discriminator[0] == 0) {
    ty = &t->types[0];
  } else {
    const Field* d = nullptr;
    for (int i = 0; i < nf; ++i)
      if (name_is(t->discriminator, f[i].key, f[i].key_len)) d = &f[i];
    if (!d || !d->str) return corrupt(std::string("event has no string field \"") + t->discriminator + "\"");
    for (uint32_t i = 0; i < t->n_types && !ty; ++i)
      if (!d->str_escaped && name_is(t->types[i].name, d->str, d->str_len)) ty = &t->types[i];
    if (!ty) return corrupt("unknown event type \"" + std::string((const char*)d->str, (size_t)d->str_len) + "\"");
  }
  auto find_num = [&](const char* name) -> const Field* {
    for (int i = 0; i < nf; ++i)
      if (f[i].num && name_is(name, f[i].key, f[i].key_len)) return &f[i];
    return nullptr;
  };
  int32_t seq = 0;
  if (ty->seq_field[0]) {
    const Field* s = find_num(ty->seq_field);
    if (!s) return corrupt(std::string("event has no numeric field \"") + ty->seq_field + "\"");
    if (parse_i32(s->num, s->num_len, &seq) != 0) return corrupt(std::string("field \"") + ty->seq_field + "\" is not an Int");
  }
  uint64_t raw = 0;
  if (ty->arg_kind != SURGE_EVJ_ARG_NONE) {
    const Field* a = find_num(ty->arg_field);
    if (!a) return corrupt(std::string("event has no numeric field \"") + ty->arg_field + "\"");
    if (ty->arg_kind == SURGE_EVJ_ARG_I32) {
      int32_t v = 0;
      if (parse_i32(a->num, a->num_len, &v) != 0) return corrupt(std::string("field \"") + ty->arg_field + "\" is not an Int");
      raw = (uint64_t)(uint32_t)v;  // arg in the low word, high word zero (surge_event16)
    } else {
      double v = 0.0;
      if (!parse_f64(a->num, a->num_len, &v)) return corrupt(std::string("field \"") + ty->arg_field + "\" is not a number");
      std::memcpy(&raw, &v, 8);
    }
  }
  uint8_t* o = (uint8_t*)event16_out;
  const int32_t type = (int32_t)ty->event_type;
  std::memcpy(o, &type, 4);
  std::memcpy(o + 4, &seq, 4);
  std::memcpy(o + 8, &raw, 8);
  return 0;
}

}  // extern "C"
