// Host range coder behind include/lic_codec.h.  Carry-propagating 32-bit range coder
// with byte-wise renormalisation (the LZMA scheme: 64-bit `low`, a cached byte and a run of 0xFF
// bytes absorb carries), 16-bit cumulative frequencies, Elias-gamma escapes in equiprobable bits.
#include <math.h>
#include <new>
#include <stddef.h>
#include <stdint.h>

#include "lic_codec.h"

#define LIC_CODEC_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

constexpr uint32_t kTop = 1u << 24;
constexpr int kBits = 16;

struct Encoder {
  uint64_t low = 0;
  uint32_t range = 0xFFFFFFFFu;
  uint8_t cache = 0;
  uint64_t cache_size = 1;
  uint8_t* out;
  size_t cap, pos = 0;
  bool overflow = false;
  Encoder(uint8_t* o, size_t c) : out(o), cap(c) {}
  void put(uint8_t b) {
    if (pos < cap)
      out[pos++] = b;
    else
      overflow = true;
  }
  void shift_low() {
    if ((uint32_t)low < 0xFF000000u || (low >> 32) != 0) {
      const uint8_t carry = (uint8_t)(low >> 32);
      uint8_t temp = cache;
      do {
        put((uint8_t)(temp + carry));
        temp = 0xFF;
      } while (--cache_size);
      cache = (uint8_t)((low >> 24) & 0xFF);
    }
    ++cache_size;
    low = (low & 0x00FFFFFFu) << 8;
  }
  void encode(uint32_t lo, uint32_t hi) {  // interval [lo, hi) of 65536
    const uint32_t r = range >> kBits;
    low += (uint64_t)r * lo;
    range = r * (hi - lo);
    while (range < kTop) {
      range <<= 8;
      shift_low();
    }
  }
  void bit(int b) { encode(b ? 32768u : 0u, b ? 65536u : 32768u); }
  void gamma(uint32_t v) {  // Elias-gamma of v + 1
    const uint64_t x = (uint64_t)v + 1;
    int nb = 0;
    while ((x >> (nb + 1)) != 0) ++nb;
    for (int i = 0; i < nb; ++i) bit(0);
    bit(1);
    for (int i = nb - 1; i >= 0; --i) bit((int)((x >> i) & 1));
  }
  void flush() {
    for (int i = 0; i < 5; ++i) shift_low();
  }
};

struct Decoder {
  uint32_t code = 0, range = 0xFFFFFFFFu;
  const uint8_t* in;
  size_t n, pos = 0;
  bool underflow = false;
  Decoder(const uint8_t* i, size_t nb) : in(i), n(nb) {
    for (int k = 0; k < 5; ++k) code = (code << 8) | next();
  }
  uint8_t next() {
    if (pos < n) return in[pos++];
    underflow = true;
    return 0;
  }
  uint32_t target() {
    const uint32_t t = code / (range >> kBits);
    return t > 65535u ? 65535u : t;
  }
  void consume(uint32_t lo, uint32_t hi) {
    const uint32_t r = range >> kBits;
    code -= r * lo;
    range = r * (hi - lo);
    while (range < kTop) {
      code = (code << 8) | next();
      range <<= 8;
    }
  }
  int bit() {
    const int b = target() >= 32768u;
    consume(b ? 32768u : 0u, b ? 65536u : 32768u);
    return b;
  }
  bool gamma(uint32_t* v) {
    int nb = 0;
    while (!bit()) {
      if (++nb > 32 || underflow) return false;
    }
    uint64_t x = 1;
    for (int i = 0; i < nb; ++i) x = (x << 1) | (uint64_t)bit();
    *v = (uint32_t)(x - 1);
    return true;
  }
};

inline bool table_ok(const uint32_t* t, int S) { return t[0] == 0 && t[S] == 65536u; }

}  // namespace

LIC_CODEC_EXPORT size_t lic_rc_bound(int64_t n) {
  // <= 2 bytes per in-window symbol (16-bit frequencies), escapes add <= 65 bits; generous bound
  return n < 0 ? 0 : (size_t)n * 12 + 64;
}

LIC_CODEC_EXPORT int lic_rc_encode(const uint32_t* tables, const int32_t* table_of, int32_t S, const int32_t* idx,
                                   int64_t n, uint8_t* out, size_t cap, size_t* nbytes) {
  if (!tables || !idx || !out || !nbytes || S < 2 || n < 0) return LIC_CODEC_ERR_INVALID;
  Encoder e(out, cap);
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t* t = tables + (size_t)(table_of ? table_of[i] : i) * (size_t)(S + 1);
    if (!table_ok(t, S)) return LIC_CODEC_ERR_INVALID;
    const int32_t v = idx[i];
    const int32_t s = v <= 0 ? 0 : (v >= S - 1 ? S - 1 : v);
    if (t[s + 1] <= t[s]) return LIC_CODEC_ERR_INVALID;
    e.encode(t[s], t[s + 1]);
    if (s == 0) e.gamma((uint32_t)(-(int64_t)v));
    if (s == S - 1) e.gamma((uint32_t)((int64_t)v - (S - 1)));
  }
  e.flush();
  if (e.overflow) return LIC_CODEC_ERR_OVERFLOW;
  *nbytes = e.pos;
  return LIC_CODEC_OK;
}

static int decode_some(Decoder& d, const uint32_t* tables, const int32_t* table_of, int32_t S, int64_t n,
                       int32_t* idx_out) {
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t* t = tables + (size_t)(table_of ? table_of[i] : i) * (size_t)(S + 1);
    if (!table_ok(t, S)) return LIC_CODEC_ERR_INVALID;
    const uint32_t tg = d.target();
    int lo = 0, hi = S;  // largest s with t[s] <= tg
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (t[mid] <= tg)
        lo = mid;
      else
        hi = mid;
    }
    d.consume(t[lo], t[lo + 1]);
    int64_t v = lo;
    if (lo == 0 || lo == S - 1) {
      uint32_t ex = 0;
      if (!d.gamma(&ex)) return LIC_CODEC_ERR_CORRUPT;
      v = lo == 0 ? -(int64_t)ex : (int64_t)(S - 1) + ex;
      if (v < INT32_MIN || v > INT32_MAX) return LIC_CODEC_ERR_CORRUPT;
    }
    idx_out[i] = (int32_t)v;
    if (d.underflow) return LIC_CODEC_ERR_CORRUPT;
  }
  return LIC_CODEC_OK;
}

LIC_CODEC_EXPORT int lic_rc_decode(const uint8_t* in, size_t nbytes, const uint32_t* tables, const int32_t* table_of,
                                   int32_t S, int64_t n, int32_t* idx_out) {
  if (!in || !tables || !idx_out || S < 2 || n < 0) return LIC_CODEC_ERR_INVALID;
  Decoder d(in, nbytes);
  return decode_some(d, tables, table_of, S, n, idx_out);
}

// streaming form: the tables of later symbols may depend on earlier decoded ones (context models)
struct lic_rc_decoder {
  Decoder d;
  lic_rc_decoder(const uint8_t* in, size_t n) : d(in, n) {}
};
LIC_CODEC_EXPORT lic_rc_decoder* lic_rc_decoder_new(const uint8_t* in, size_t nbytes) {
  if (!in) return nullptr;
  return new (std::nothrow) lic_rc_decoder(in, nbytes);
}
LIC_CODEC_EXPORT int lic_rc_decoder_next(lic_rc_decoder* dec, const uint32_t* tables, const int32_t* table_of,
                                         int32_t S, int64_t n, int32_t* idx_out) {
  if (!dec || !tables || !idx_out || S < 2 || n < 0) return LIC_CODEC_ERR_INVALID;
  return decode_some(dec->d, tables, table_of, S, n, idx_out);
}
LIC_CODEC_EXPORT void lic_rc_decoder_free(lic_rc_decoder* dec) { delete dec; }

LIC_CODEC_EXPORT double lic_rc_ideal_bits(const uint32_t* tables, const int32_t* table_of, int32_t S,
                                          const int32_t* idx, int64_t n) {
  if (!tables || !idx || S < 2 || n < 0) return -1.0;
  double bits = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t* t = tables + (size_t)(table_of ? table_of[i] : i) * (size_t)(S + 1);
    const int32_t v = idx[i];
    const int32_t s = v <= 0 ? 0 : (v >= S - 1 ? S - 1 : v);
    bits += 16.0 - log2((double)(t[s + 1] - t[s]));
    if (s == 0 || s == S - 1) {
      const uint64_t x = (uint64_t)(s == 0 ? -(int64_t)v : (int64_t)v - (S - 1)) + 1;
      int nb = 0;
      while ((x >> (nb + 1)) != 0) ++nb;
      bits += 2 * nb + 1;
    }
  }
  return bits;
}

LIC_CODEC_EXPORT int lic_codec_version(void) { return 1; }
