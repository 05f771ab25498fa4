// f64_text.h — the JSON text play-json writes for a Scala Double, on the host and on the device (same code).
//
// What the reference does with a Double field of a state or an event (BankAccount.balance,
// modules/surge-docs/src/test/scala/docs/command/BankAccountSurgeModel.scala:26-28: Json.toJson(agg).toString()), through
// play-json 2.9.2 (project/Dependencies.scala:73) — a dependency that is not vendored under /root/reference, restated
// here from its published sources:
//   1. Writes.DoubleWrites: JsNumber(BigDecimal(d)) = new java.math.BigDecimal(java.lang.Double.toString(d)) under
//      MathContext.DECIMAL128, i.e. the decimal digits of Double.toString — the SHORTEST decimal that rounds back to d,
//      the closest one when several are equally short, and — Double.toString always prints a fraction digit — when the
//      shortest has ONE digit, the closest of the TWO-digit decimals that round back (java.lang.Double.toString's
//      specification since JDK 19; it only changes the value for a few hundred subnormals: Double.MIN_VALUE is 4.9E-324,
//      not 5E-324).  JDK >= 19 implements exactly that; older JDKs print one digit too many for a small set of values
//      (JDK-4511638), which is outside what can be pinned without the JVM that wrote the topic.  NaN / infinities have
//      no BigDecimal: the reference throws.
//   2. JsValueSerializer (play-json JacksonJson.scala): stripped = v.stripTrailingZeros; 1E-10 < |v| < 1E20 ?
//      stripped.toPlainString : stripped.toString; the text is re-read as a BigDecimal (or BigInteger when it has no '.'
//      / 'E') and Jackson writes that number's toString.
//   Net effect, for digits d1 d2 .. dn (no trailing zeros) and adjusted exponent a (|v| = d1.d2..dn x 10^a):
//      v == 0 (either sign)      0
//      a >= 20                   d1[.d2..dn]E+a          scientific, explicit '+'
//      0 <= a < 20, n <= a + 1   d1..dn followed by a + 1 - n zeros        (an integer: no ".0")
//      0 <= a < 20, n >  a + 1   d1..d(a+1) . d(a+2)..dn
//      -6 <= a < 0               0. (-a - 1 zeros) d1..dn
//      a < -6                    d1[.d2..dn]E-|a|        BigDecimal.toString's scientific form
//   with a leading '-' for negative values.
// The shortest digits come from Ryu (Ulf Adams, "Ryu: fast float-to-string conversion", PLDI 2018): one 64 x 128-bit
// multiplication by a tabulated power of 5 per bound, then digit removal inside the rounding interval.  The two tables
// (5^i and 2^k / 5^i to 125 bits) are computed at load time with exact integer arithmetic (f64_text.cpp).
#pragma once
#include <stdint.h>

namespace surge {

constexpr int kPow5InvBitCount = 125;
constexpr int kPow5BitCount = 125;
constexpr int kPow5InvTableSize = 342;
constexpr int kPow5TableSize = 326;

struct F64Tables {
  uint64_t pow5_inv[kPow5InvTableSize][2];  // floor(2^(pow5bits(i) - 1 + 125) / 5^i) + 1, {low, high}
  uint64_t pow5[kPow5TableSize][2];         // 5^i scaled to 125 bits: 5^i >> (pow5bits(i) - 125)
};

const F64Tables* f64_tables_host();  // f64_text.cpp: built on first use

#if defined(__HIPCC__)
#define SURGE_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define SURGE_HD inline
#endif

SURGE_HD uint32_t ryu_pow5bits(int32_t e) { return (uint32_t)(((uint32_t)e * 1217359u) >> 19) + 1u; }   // ceil(log2(5^e)), 0 <= e <= 3528
SURGE_HD uint32_t ryu_log10pow2(int32_t e) { return ((uint32_t)e * 78913u) >> 18; }                    // floor(log10(2^e)), 0 <= e <= 1650
SURGE_HD uint32_t ryu_log10pow5(int32_t e) { return ((uint32_t)e * 732923u) >> 20; }                   // floor(log10(5^e)), 0 <= e <= 2620

SURGE_HD uint64_t ryu_umul128(uint64_t a, uint64_t b, uint64_t* hi) {
  const unsigned __int128 p = (unsigned __int128)a * b;  // host: one mul; gfx950: 32-bit partial products (v_mad_u64_u32)
  *hi = (uint64_t)(p >> 64);
  return (uint64_t)p;
}

// (m * mul) >> j for a 128-bit mul = {low, high}, 64 < j < 128; m has at most 55 bits
SURGE_HD uint64_t ryu_mulshift64(uint64_t m, const uint64_t* mul, int32_t j) {
  uint64_t high1, high0;
  const uint64_t low1 = ryu_umul128(m, mul[1], &high1);
  (void)ryu_umul128(m, mul[0], &high0);
  const uint64_t sum = high0 + low1;
  if (sum < high0) ++high1;
  const int32_t dist = j - 64;
  return (high1 << (64 - dist)) | (sum >> dist);
}

SURGE_HD uint32_t ryu_pow5factor(uint64_t v) {
  uint32_t c = 0;
  for (;;) {
    const uint64_t q = v / 5u;
    if (v - 5u * q != 0u) break;
    v = q;
    ++c;
  }
  return c;
}

// Shortest decimal of a finite, non-zero double given by its IEEE fields: value = *digits x 10^(*exp10), *digits
// without trailing zeros (1..17 digits).
SURGE_HD void ryu_shortest(uint64_t ieee_mantissa, uint32_t ieee_exponent, const F64Tables* tb, uint64_t* digits, int32_t* exp10) {
  int32_t e2;
  uint64_t m2;
  if (ieee_exponent == 0u) {
    e2 = 1 - 1023 - 52 - 2;
    m2 = ieee_mantissa;
  } else {
    e2 = (int32_t)ieee_exponent - 1023 - 52 - 2;
    m2 = (1ull << 52) | ieee_mantissa;
  }
  const bool accept_bounds = (m2 & 1u) == 0u;  // round-half-even: the interval's ends round back iff the mantissa is even
  const uint64_t mv = 4u * m2;
  const uint32_t mm_shift = (ieee_mantissa != 0u || ieee_exponent <= 1u) ? 1u : 0u;  // 0: the gap below is half the gap above
  uint64_t vr, vp, vm;
  int32_t e10;
  bool vm_tz = false, vr_tz = false;
  if (e2 >= 0) {
    const uint32_t q = ryu_log10pow2(e2) - (e2 > 3 ? 1u : 0u);
    e10 = (int32_t)q;
    const int32_t k = kPow5InvBitCount + (int32_t)ryu_pow5bits((int32_t)q) - 1;
    const int32_t i = -e2 + (int32_t)q + k;
    const uint64_t* mul = tb->pow5_inv[q];
    vr = ryu_mulshift64(4u * m2, mul, i);
    vp = ryu_mulshift64(4u * m2 + 2u, mul, i);
    vm = ryu_mulshift64(4u * m2 - 1u - mm_shift, mul, i);
    if (q <= 21u) {
      const uint32_t mv_mod5 = (uint32_t)(mv - 5u * (mv / 5u));
      if (mv_mod5 == 0u) {
        vr_tz = ryu_pow5factor(mv) >= q;
      } else if (accept_bounds) {
        vm_tz = ryu_pow5factor(mv - 1u - mm_shift) >= q;
      } else {
        vp -= ryu_pow5factor(mv + 2u) >= q ? 1u : 0u;
      }
    }
  } else {
    const uint32_t q = ryu_log10pow5(-e2) - (-e2 > 1 ? 1u : 0u);
    e10 = (int32_t)q + e2;
    const int32_t i = -e2 - (int32_t)q;
    const int32_t k = (int32_t)ryu_pow5bits(i) - kPow5BitCount;
    const int32_t j = (int32_t)q - k;
    const uint64_t* mul = tb->pow5[i];
    vr = ryu_mulshift64(4u * m2, mul, j);
    vp = ryu_mulshift64(4u * m2 + 2u, mul, j);
    vm = ryu_mulshift64(4u * m2 - 1u - mm_shift, mul, j);
    if (q <= 1u) {
      vr_tz = true;  // mv = 4 m2 has at least two trailing zero bits
      if (accept_bounds) {
        vm_tz = mm_shift == 1u;
      } else {
        --vp;
      }
    } else if (q < 63u) {
      vr_tz = (mv & ((1ull << q) - 1ull)) == 0ull;
    }
  }
  // remove digits while the interval still contains more than one candidate; remember what was cut from vr to round
  int32_t removed = 0;
  uint32_t last_removed = 0u;
  for (;;) {
    const uint64_t vp10 = vp / 10u, vm10 = vm / 10u;
    if (vp10 <= vm10) break;
    const uint32_t vm_mod = (uint32_t)(vm - 10u * vm10);
    const uint64_t vr10 = vr / 10u;
    const uint32_t vr_mod = (uint32_t)(vr - 10u * vr10);
    vm_tz = vm_tz && vm_mod == 0u;
    vr_tz = vr_tz && last_removed == 0u;
    last_removed = vr_mod;
    vr = vr10; vp = vp10; vm = vm10;
    ++removed;
  }
  if (vm_tz) {
    for (;;) {
      const uint64_t vm10 = vm / 10u;
      const uint32_t vm_mod = (uint32_t)(vm - 10u * vm10);
      if (vm_mod != 0u) break;
      const uint64_t vp10 = vp / 10u, vr10 = vr / 10u;
      const uint32_t vr_mod = (uint32_t)(vr - 10u * vr10);
      vr_tz = vr_tz && last_removed == 0u;
      last_removed = vr_mod;
      vr = vr10; vp = vp10; vm = vm10;
      ++removed;
    }
  }
  if (vr_tz && last_removed == 5u && (vr & 1u) == 0u) last_removed = 4u;  // exactly ...50..0: round half to even
  uint64_t out = vr + (((vr == vm && (!accept_bounds || !vm_tz)) || last_removed >= 5u) ? 1u : 0u);
  int32_t e = e10 + removed;
  for (;;) {  // small integers and exact powers of ten leave zeros at the end: move them into the exponent
    const uint64_t q = out / 10u;
    if (out - 10u * q != 0u) break;
    out = q;
    ++e;
  }
  *digits = out;
  *exp10 = e;
}

constexpr int kF64TextMax = 26;  // '-' + 17 digits + '.' + "E-324", or "0.00000" + 17 digits

// The text described at the top for the double with these bits; returns its length, 0 for NaN / infinity (no JSON
// number exists: the reference's writeState throws).  out (nullable: length only) must hold kF64TextMax bytes.
SURGE_HD int f64_play_json_text(uint64_t bits, const F64Tables* tb, uint8_t* out) {
  const uint64_t mant = bits & ((1ull << 52) - 1ull);
  const uint32_t expo = (uint32_t)((bits >> 52) & 0x7ffu);
  if (expo == 0x7ffu) return 0;
  if (expo == 0u && mant == 0ull) {
    if (out) out[0] = '0';
    return 1;
  }
  uint64_t digits;
  int32_t e10;
  ryu_shortest(mant, expo, tb, &digits, &e10);
  if (digits < 10u && expo == 0u && mant <= 1000u) {
    // A one-digit shortest decimal: Double.toString prints the closest TWO-digit decimal instead.  For normal doubles
    // that is the same number (d.0: the rounding interval is far narrower than the spacing of two-digit decimals); only
    // the smallest subnormals have room for another one.  x = mant * 2^-1074 = mant * 4.9406564584124654...e-324: a
    // double product decides the two digits for every mant <= 1000 (no product comes within 1e-4 of a rounding
    // midpoint; tests/test_f64_text.py checks all of them against exact rational arithmetic).
    const double v = (double)mant * 4.9406564584124654;  // x * 1e324
    double scaled = v;
    int32_t e = -324;                                     // exponent of the units digit of `scaled`
    if (v < 10.0) { scaled = v * 10.0; e = -325; }
    else if (v >= 1000.0) { scaled = v / 100.0; e = -322; }
    else if (v >= 100.0) { scaled = v / 10.0; e = -323; }
    uint64_t two = (uint64_t)(scaled + 0.5);
    if (two == 100u) { two = 10u; ++e; }
    digits = two;
    e10 = e;
    if (digits % 10u == 0u) { digits /= 10u; ++e10; }
  }
  uint8_t d[20];
  int n = 0;
  for (uint64_t v = digits; v != 0ull; v /= 10u) d[n++] = (uint8_t)('0' + (int)(v % 10u));  // least significant first
  const int a = e10 + n - 1;  // adjusted exponent
  int len = 0;
  auto put = [&](uint8_t c) {
    if (out) out[len] = c;
    ++len;
  };
  if (bits >> 63) put('-');
  if (a >= 20 || a < -6) {
    put(d[n - 1]);
    if (n > 1) {
      put('.');
      for (int i = n - 2; i >= 0; --i) put(d[i]);
    }
    put('E');
    put(a < 0 ? '-' : '+');
    uint32_t x = (uint32_t)(a < 0 ? -a : a);
    if (x >= 100u) put((uint8_t)('0' + x / 100u));
    if (x >= 10u) put((uint8_t)('0' + (x / 10u) % 10u));
    put((uint8_t)('0' + x % 10u));
  } else if (a >= 0) {
    for (int i = 0; i <= a; ++i) put(i < n ? d[n - 1 - i] : (uint8_t)'0');
    if (n > a + 1) {
      put('.');
      for (int i = n - 2 - a; i >= 0; --i) put(d[i]);
    }
  } else {
    put('0');
    put('.');
    for (int i = 0; i < -a - 1; ++i) put('0');
    for (int i = n - 1; i >= 0; --i) put(d[i]);
  }
  return len;
}

}  // namespace surge
