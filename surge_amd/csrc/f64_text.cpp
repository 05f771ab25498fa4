// f64_text.cpp — the power-of-5 tables of f64_text.h, computed once with exact integer arithmetic, and the host entry
// point of the Double -> JSON text conversion (the device copy of the tables is made by engine.hip).
#include "f64_text.h"
#include "f64_parse.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/surge_replay.h"

namespace surge {
namespace {

// little-endian base-2^32 naturals, just enough arithmetic for 5^i (i <= 341: 792 bits) and 2^k / 5^i
typedef std::vector<uint32_t> Nat;

void mul_small(Nat& a, uint32_t m) {
  uint64_t carry = 0;
  for (uint32_t& limb : a) {
    const uint64_t v = (uint64_t)limb * m + carry;
    limb = (uint32_t)v;
    carry = v >> 32;
  }
  if (carry) a.push_back((uint32_t)carry);
}

int bit_length(const Nat& a) {
  for (int i = (int)a.size() - 1; i >= 0; --i)
    if (a[i]) return i * 32 + (32 - __builtin_clz(a[i]));
  return 0;
}

bool bit_at(const Nat& a, int b) { return b >= 0 && (size_t)(b / 32) < a.size() && ((a[b / 32] >> (b % 32)) & 1u); }

// bits [lo, lo + 128) of a (lo may be negative: shifted left)
void window128(const Nat& a, int lo, uint64_t out[2]) {
  out[0] = out[1] = 0;
  for (int b = 0; b < 128; ++b)
    if (bit_at(a, lo + b)) out[b / 64] |= 1ull << (b % 64);
}

// floor(2^k / d) by schoolbook long division, one quotient bit at a time (k <= 920, d <= 800 bits: a few ms in total)
Nat div_pow2(int k, const Nat& d) {
  Nat q((size_t)(k / 32 + 1), 0u), r;
  auto shl1_add = [](Nat& x, uint32_t bit) {
    uint32_t carry = bit;
    for (uint32_t& limb : x) {
      const uint32_t nc = limb >> 31;
      limb = (limb << 1) | carry;
      carry = nc;
    }
    if (carry) x.push_back(carry);
  };
  auto geq = [](const Nat& x, const Nat& y) {
    const size_t n = x.size() > y.size() ? x.size() : y.size();
    for (size_t i = n; i-- > 0;) {
      const uint32_t a = i < x.size() ? x[i] : 0u, b = i < y.size() ? y[i] : 0u;
      if (a != b) return a > b;
    }
    return true;
  };
  auto sub = [](Nat& x, const Nat& y) {
    int64_t borrow = 0;
    for (size_t i = 0; i < x.size(); ++i) {
      int64_t v = (int64_t)x[i] - (i < y.size() ? (int64_t)y[i] : 0) - borrow;
      borrow = v < 0;
      if (v < 0) v += (int64_t)1 << 32;
      x[i] = (uint32_t)v;
    }
  };
  for (int b = k; b >= 0; --b) {
    shl1_add(r, b == k ? 1u : 0u);  // the dividend is 1 followed by k zeros
    if (geq(r, d)) {
      sub(r, d);
      q[b / 32] |= 1u << (b % 32);
    }
  }
  return q;
}

F64Tables g_tables;
std::once_flag g_once;

void build_tables() {
  Nat p{1u};  // 5^i
  const int n = kPow5InvTableSize > kPow5TableSize ? kPow5InvTableSize : kPow5TableSize;
  for (int i = 0; i < n; ++i) {
    const int len = bit_length(p);  // == ryu_pow5bits(i) (checked by the tests through the conversion's results)
    if (i < kPow5TableSize) window128(p, len - kPow5BitCount, g_tables.pow5[i]);
    if (i < kPow5InvTableSize) {
      Nat q = div_pow2(len - 1 + kPow5InvBitCount, p);
      // + 1
      for (size_t l = 0; l < q.size(); ++l)
        if (++q[l] != 0u) break;
      window128(q, 0, g_tables.pow5_inv[i]);
    }
    mul_small(p, 5u);
  }
}

// The Eisel-Lemire table (f64_parse.h), as the algorithm's published generator script defines it:
//   q >= 0: 5^q shifted left until bit 127 is set, truncated to 128 bits;
//   q <  0: c = floor(2^b / 5^-q) + 1 with b = z + 127 for q >= -27 and b = 2 z + 128 below (z = bits of 5^-q, i.e. the
//           smallest z with 2^z >= 5^-q), truncated to its top 128 bits.
F64ParseTable g_parse;
std::once_flag g_parse_once;

void build_parse_table() {
  auto store = [](int q, const Nat& v) {
    uint64_t w[2];
    window128(v, bit_length(v) - 128, w);  // top 128 bits (shifted left when shorter)
    g_parse.p5[2 * (q - kPow10Min)] = w[1];
    g_parse.p5[2 * (q - kPow10Min) + 1] = w[0];
  };
  Nat p{1u};
  for (int q = 0; q <= kPow10Max; ++q) {
    store(q, p);
    mul_small(p, 5u);
  }
  p = Nat{5u};
  for (int q = -1; q >= kPow10Min; --q) {
    int z = bit_length(p);                       // 2^(z-1) <= p < 2^z; p is never a power of two, so this is the script's z
    const int b = q >= -27 ? z + 127 : 2 * z + 128;
    Nat c = div_pow2(b, p);
    for (size_t l = 0;; ++l) {                   // + 1
      if (l == c.size()) { c.push_back(1u); break; }
      if (++c[l] != 0u) break;
    }
    store(q, c);
    mul_small(p, 5u);
  }
}

}  // namespace

const F64ParseTable* f64_parse_table_host() {
  std::call_once(g_parse_once, build_parse_table);
  return &g_parse;
}

const F64Tables* f64_tables_host() {
  std::call_once(g_once, build_tables);
  return &g_tables;
}

}  // namespace surge

extern "C" int32_t surge_format_f64_json(uint64_t bits, uint8_t* out, int32_t capacity) {
  uint8_t tmp[surge::kF64TextMax];
  const int n = surge::f64_play_json_text(bits, surge::f64_tables_host(), tmp);
  if (out && capacity >= n) std::memcpy(out, tmp, (size_t)n);
  return n;
}

// JSON number -> double bits.  0 = OK; 1 = the fast algorithm cannot decide (more than 19 digits, or one of its rare
// ambiguous products): *bits_out is then the exact result computed with strtod — the status only tells tests which path
// ran; SURGE_E_CORRUPT (-7) = not a JSON number.
extern "C" int32_t surge_parse_f64_json(const uint8_t* text, int64_t len, uint64_t* bits_out) {
  if (!text || !bits_out || len <= 0 || len >= 400) return -7;
  const int rc = surge::f64_parse_json_number(text, (int)len, surge::f64_parse_table_host(), bits_out);
  if (rc == surge::F64_PARSE_MALFORMED) return -7;
  if (rc == surge::F64_PARSE_AMBIGUOUS) {
    char buf[400];
    std::memcpy(buf, text, (size_t)len);
    buf[len] = 0;
    char* endp = nullptr;
    const double v = std::strtod(buf, &endp);
    if (endp != buf + len) return -7;
    std::memcpy(bits_out, &v, 8);
    return 1;
  }
  return 0;
}

extern "C" int64_t surge_format_f64_json_many(const uint64_t* bits, int64_t n, uint8_t* out, int64_t capacity, int64_t* out_off) {
  const surge::F64Tables* tb = surge::f64_tables_host();
  int64_t pos = 0;
  for (int64_t i = 0; i < n; ++i) {
    uint8_t tmp[surge::kF64TextMax];
    const int len = surge::f64_play_json_text(bits[i], tb, tmp);
    if (out_off) out_off[i] = pos;
    if (out && pos + len <= capacity) std::memcpy(out + pos, tmp, (size_t)len);
    pos += len;
  }
  if (out_off) out_off[n] = pos;
  return pos;
}
