// f64_parse.h — a JSON number -> IEEE double, correctly rounded, on the host and on the device (same code): what
// play-json does when it reads a Double field of an event (BankAccountCreated.balance / BankAccountUpdated.newBalance,
// modules/surge-docs/src/test/scala/docs/command/BankAccountCommandModel.scala:39-47) — BigDecimal(text).doubleValue, i.e.
// the double nearest to the decimal, ties to even.
//
// Method: the Eisel–Lemire algorithm (D. Lemire, "Number parsing at a gigabyte per second", SPE 2021; the algorithm of
// fast_float / Go's strconv / Rust's core): decimal significand w (up to 19 digits, exact in 64 bits) and exponent q ->
// one or two 64 x 64-bit multiplications by a 128-bit approximation of 5^q -> the 53-bit mantissa with a proof that the
// truncated product decides the rounding — except in rare cases it detects and REPORTS (status F64_PARSE_AMBIGUOUS:
// the caller re-parses that value with an exact method; the device decoder hands those records back to the host).  The
// 651-entry table (5^q for q in [-342, 308], 128 bits, rounded as the published generator script does) is built at load
// time with exact integer arithmetic (f64_text.cpp).
#pragma once
#include <stdint.h>

#include "f64_text.h"  // SURGE_HD

namespace surge {

constexpr int kPow10Min = -342, kPow10Max = 308;
struct F64ParseTable {
  uint64_t p5[(kPow10Max - kPow10Min + 1) * 2];  // {high, low} of the 128-bit 5^q, most significant bit set
};
const F64ParseTable* f64_parse_table_host();  // f64_text.cpp: built on first use

enum { F64_PARSE_OK = 0, F64_PARSE_AMBIGUOUS = 1, F64_PARSE_MALFORMED = 2 };

SURGE_HD int f64_clz64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_clzll(v);
#else
  return __builtin_clzll(v);
#endif
}

// w x 10^q -> double bits (sign applied by the caller).  Returns F64_PARSE_OK or F64_PARSE_AMBIGUOUS.
SURGE_HD int f64_from_decimal(uint64_t w, int64_t q, const F64ParseTable* tb, uint64_t* bits) {
  constexpr int kMantBits = 52, kMinExp = -1023, kInfPower = 0x7FF;
  if (w == 0 || q < kPow10Min) { *bits = 0; return F64_PARSE_OK; }
  if (q > kPow10Max) { *bits = (uint64_t)kInfPower << 52; return F64_PARSE_OK; }
  const int lz = f64_clz64(w);
  w <<= lz;
  // 128-bit product approximation with mantissa bits + 3 bits of precision
  const int index = 2 * (int)(q - kPow10Min);
  uint64_t hi, lo;
  lo = ryu_umul128(w, tb->p5[index], &hi);
  constexpr uint64_t precision_mask = 0xFFFFFFFFFFFFFFFFull >> (kMantBits + 3);
  if ((hi & precision_mask) == precision_mask) {
    uint64_t hi2;
    (void)ryu_umul128(w, tb->p5[index + 1], &hi2);
    lo += hi2;
    if (hi2 > lo) ++hi;
  }
  if (lo == 0xFFFFFFFFFFFFFFFFull) {
    const bool inside_safe_exponent = q >= -27 && q <= 55;
    if (!inside_safe_exponent) return F64_PARSE_AMBIGUOUS;
  }
  const int upperbit = (int)(hi >> 63);
  uint64_t mantissa = hi >> (upperbit + 64 - kMantBits - 3);
  int32_t power2 = (int32_t)((((152170 + 65536) * (int32_t)q) >> 16) + 63) + upperbit - lz - kMinExp;
  if (power2 <= 0) {  // subnormal (or zero)
    if (-power2 + 1 >= 64) { *bits = 0; return F64_PARSE_OK; }
    mantissa >>= -power2 + 1;
    mantissa += (mantissa & 1);
    mantissa >>= 1;
    power2 = (mantissa < (1ull << kMantBits)) ? 0 : 1;
    *bits = ((uint64_t)power2 << 52) | (mantissa & ((1ull << kMantBits) - 1));
    return F64_PARSE_OK;
  }
  // round to even when the decimal sits exactly between two doubles (only possible when 5^q fits 64 bits)
  if (lo <= 1 && q >= -4 && q <= 23 && (mantissa & 3) == 1) {
    if ((mantissa << (upperbit + 64 - kMantBits - 3)) == hi) mantissa &= ~1ull;
  }
  mantissa += (mantissa & 1);
  mantissa >>= 1;
  if (mantissa >= (2ull << kMantBits)) {
    mantissa = 1ull << kMantBits;
    ++power2;
  }
  mantissa &= ~(1ull << kMantBits);
  if (power2 >= kInfPower) { *bits = (uint64_t)kInfPower << 52; return F64_PARSE_OK; }
  *bits = ((uint64_t)power2 << 52) | mantissa;
  return F64_PARSE_OK;
}

// A JSON number literal s[0..len) (RFC 8259 grammar: -? int frac? exp?; a leading '+' is tolerated as the host decoder's
// strtod tolerates it) -> double bits.  More than 19 significant digits: AMBIGUOUS (the exact method decides).
SURGE_HD int f64_parse_json_number(const uint8_t* s, int len, const F64ParseTable* tb, uint64_t* bits) {
  int i = 0;
  bool neg = false;
  if (i < len && (s[i] == '-' || s[i] == '+')) { neg = s[i] == '-'; ++i; }
  uint64_t w = 0;
  int digits = 0;       // significant digits taken into w
  int64_t exp10 = 0;    // decimal exponent of w's last digit
  bool any = false, dropped_nonzero = false;
  while (i < len && s[i] >= '0' && s[i] <= '9') {
    any = true;
    const uint32_t d = (uint32_t)(s[i] - '0');
    if (digits < 19) {
      if (w != 0 || d != 0) { w = w * 10 + d; ++digits; }
    } else {
      ++exp10;
      dropped_nonzero = dropped_nonzero || d != 0;
    }
    ++i;
  }
  if (i < len && s[i] == '.') {
    ++i;
    bool frac = false;
    while (i < len && s[i] >= '0' && s[i] <= '9') {
      frac = true;
      const uint32_t d = (uint32_t)(s[i] - '0');
      if (digits < 19) {
        if (w != 0 || d != 0) { w = w * 10 + d; ++digits; }
        --exp10;
      } else {
        dropped_nonzero = dropped_nonzero || d != 0;
      }
      ++i;
    }
    if (!frac) return F64_PARSE_MALFORMED;
  }
  if (!any) return F64_PARSE_MALFORMED;
  if (i < len && (s[i] == 'e' || s[i] == 'E')) {
    ++i;
    bool eneg = false;
    if (i < len && (s[i] == '-' || s[i] == '+')) { eneg = s[i] == '-'; ++i; }
    int64_t e = 0;
    bool edig = false;
    while (i < len && s[i] >= '0' && s[i] <= '9') {
      edig = true;
      if (e < 100000) e = e * 10 + (s[i] - '0');
      ++i;
    }
    if (!edig) return F64_PARSE_MALFORMED;
    exp10 += eneg ? -e : e;
  }
  if (i != len) return F64_PARSE_MALFORMED;
  if (dropped_nonzero) return F64_PARSE_AMBIGUOUS;
  uint64_t b = 0;
  const int rc = f64_from_decimal(w, exp10, tb, &b);
  if (rc != F64_PARSE_OK) return rc;
  *bits = b | ((uint64_t)(neg ? 1 : 0) << 63);
  return F64_PARSE_OK;
}

}  // namespace surge
