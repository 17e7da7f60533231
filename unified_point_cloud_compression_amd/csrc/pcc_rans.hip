// rANS entropy coder: the coder CompressAI's `_CXX` extension provides to the reference at
// model/entropy_models.py:371-372,397-400,438,471,484 -- 64-bit state, 32-bit renormalisation words, 16-bit
// probabilities, 4-bit bypass digits for values outside a table (ryg_rans rans64 scheme).
//
// One implementation, two drivers:
//   * host, single stream: byte layout of `BufferedRansEncoder.flush()` / `RansDecoder.decode_with_indexes`;
//   * GPU, one stream per channel (lane = channel): the reference codes [1, C, N] tensors channel-major, so its
//     symbol order splits into C independent sub-sequences; each lane codes one of them, reads are coalesced across
//     lanes, and the streams are packed behind a small length table.  1.9 M sequential symbols become 128 x 14.8 k.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "pcc_common.h"

#define RANS_HD __host__ __device__ inline

static constexpr int R_PREC = 16;
static constexpr int R_BYP = 4;
static constexpr int R_MAXB = (1 << R_BYP) - 1;
static constexpr unsigned long long R_L = 1ull << 31;

struct RansTab { const int* cdf; int stride; const int* sizes; const int* offsets; };

// Per (table row, value) division-free encoder entry (ryg_rans `Rans64EncSymbol`): x -> x + bias + q*cmpl_freq with
// q = mulhi(x, rcp_freq) >> rcp_shift equals ((x / freq) << 16) + (x % freq) + start exactly.
struct RansEncSym { unsigned long long rcp_freq; unsigned bias; unsigned short cmpl_freq; unsigned short rcp_shift; };

RANS_HD unsigned long long r_mulhi(unsigned long long a, unsigned long long b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (unsigned long long)(((unsigned __int128)a * b) >> 64);
#endif
}

RANS_HD void r_put_sym(unsigned long long& x, unsigned*& ptr, const RansEncSym& e) {
  const unsigned freq = (1u << R_PREC) - e.cmpl_freq;
  const unsigned long long x_max = ((R_L >> R_PREC) << 32) * freq;
  if (x >= x_max) { *--ptr = (unsigned)x; x >>= 32; }
  const unsigned long long q = r_mulhi(x, e.rcp_freq) >> e.rcp_shift;
  x = x + e.bias + q * e.cmpl_freq;
}

// x / f and x % f for f < 2^16 and x < 2^47 * f (always true after renormalisation): three 32-bit divisions
// instead of the 64-bit software division routine.
RANS_HD void r_divmod(unsigned long long x, unsigned f, unsigned long long& q, unsigned& r) {
  const unsigned hi = (unsigned)(x >> 32), lo = (unsigned)x;
  const unsigned q0 = hi / f, r0 = hi - q0 * f;
  const unsigned d1 = (r0 << 16) | (lo >> 16);
  const unsigned q1 = d1 / f, r1 = d1 - q1 * f;
  const unsigned d2 = (r1 << 16) | (lo & 0xFFFFu);
  const unsigned q2 = d2 / f;
  r = d2 - q2 * f;
  q = ((unsigned long long)q0 << 32) | ((unsigned long long)q1 << 16) | q2;
}

RANS_HD void r_put(unsigned long long& x, unsigned*& ptr, unsigned start, unsigned freq) {
  const unsigned long long x_max = ((R_L >> R_PREC) << 32) * freq;
  if (x >= x_max) { *--ptr = (unsigned)x; x >>= 32; }
  unsigned long long q;
  unsigned r;
  r_divmod(x, freq, q, r);
  x = (q << R_PREC) + r + start;
}
RANS_HD void r_put_bits(unsigned long long& x, unsigned*& ptr, unsigned val) {
  const unsigned long long x_max = ((R_L >> 16) << 32) * (1u << (16 - R_BYP));
  if (x >= x_max) { *--ptr = (unsigned)x; x >>= 32; }
  x = (x << R_BYP) | val;
}

// encodes n symbols (read with stride) in reverse; words are written backwards from `end`; returns the first word.
// Symbols are taken in batches of RB: all table look-ups of a batch (independent of the coder state) are issued
// first so their latencies overlap, then the state updates run from registers.
static constexpr int RB = 16;

RANS_HD void r_put_bypass(unsigned long long& x, unsigned*& ptr, unsigned raw) {
  // reverse of: count digits (15, 15, ..., rest), raw digits j = 0..nb-1   (the main symbol follows, by the caller)
  int nb = 0;
  while (nb < 8 && (raw >> (nb * R_BYP)) != 0) ++nb;
  for (int j = nb - 1; j >= 0; --j) r_put_bits(x, ptr, (raw >> (j * R_BYP)) & R_MAXB);
  int val = nb, n15 = 0;
  while (val >= R_MAXB) { val -= R_MAXB; ++n15; }
  r_put_bits(x, ptr, (unsigned)val);
  for (int j = 0; j < n15; ++j) r_put_bits(x, ptr, R_MAXB);
}

// Symbol j of a stream sits at (j >> gl) * stride + (j & (2^gl - 1)) relative to the stream's first element: the
// stream covers 2^gl adjacent channels of a row-major [rows, stride] matrix, row by row (gl = 0: one column).
// Table row of symbol j: idx[same address], or ch0 + (j & (2^gl - 1)) when idx is NULL (factorised prior).
RANS_HD unsigned* r_encode(const int* sym, const int* idx, int ch0, long long n, int gl, long long stride, RansTab t,
                           const RansEncSym* enc, unsigned* end) {
  unsigned long long x = R_L;
  unsigned* ptr = end;
  const long long gm = (1ll << gl) - 1;
  for (long long base = n; base > 0; base -= RB) {
    int value[RB], ci[RB];
    unsigned raw[RB], start[RB], freq[RB];
    RansEncSym es[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const long long i = base - 1 - u;
      const long long a = (i >> gl) * stride + (i & gm);
      ci[u] = (i >= 0) ? (idx ? idx[a] : ch0 + (int)(i & gm)) : -1;
      value[u] = (i >= 0) ? sym[a] : 0;
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      raw[u] = 0xFFFFFFFFu;                                   // marker: not bypassed
      if (ci[u] >= 0) {
        const int max_value = t.sizes[ci[u]] - 2;
        int v = value[u] - t.offsets[ci[u]];
        if (v < 0) { raw[u] = (unsigned)(-2 * v - 1); v = max_value; }
        else if (v >= max_value) { raw[u] = (unsigned)(2 * (v - max_value)); v = max_value; }
        value[u] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      if (ci[u] >= 0) {
        if (enc) es[u] = enc[(long long)ci[u] * t.stride + value[u]];
        else {
          const int* c = t.cdf + (long long)ci[u] * t.stride;
          start[u] = (unsigned)c[value[u]];
          freq[u] = (unsigned)c[value[u] + 1] - start[u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      if (ci[u] >= 0) {
        if (raw[u] != 0xFFFFFFFFu) r_put_bypass(x, ptr, raw[u]);
        if (enc) r_put_sym(x, ptr, es[u]); else r_put(x, ptr, start[u], freq[u]);
      }
    }
  }
  ptr -= 2;
  ptr[0] = (unsigned)x;
  ptr[1] = (unsigned)(x >> 32);
  return ptr;
}

// The decoder keeps the next 32-bit word in a register (`nw`), re-loading right after it is consumed, so the
// renormalisation read is off the state's critical path.
// `last`: the last word that may be read (the buffer's final word).  A well-formed stream never gets there; a corrupt one
// re-reads it instead of running past the buffer, and the caller sees from the returned pointer that more was consumed than
// the stream holds.
RANS_HD unsigned r_next(const unsigned*& ptr, const unsigned* last) { ++ptr; return *(ptr <= last ? ptr : last); }

RANS_HD unsigned r_get_bits(unsigned long long& x, const unsigned*& ptr, unsigned& nw, const unsigned* last) {
  const unsigned v = (unsigned)(x & R_MAXB);
  x >>= R_BYP;
  if (x < R_L) { x = (x << 32) | nw; nw = r_next(ptr, last); }
  return v;
}

// decodes n symbols forward; returns the pointer past the last word consumed.  `d` (nullable pointers) is the
// compact decoder table: 16-bit CDF rows back to back (row_off), plus per row a 256-entry bucket table
// lut[row*256 + (cum >> 8)] = last s with cdf[s] <= (cum & ~255).  Small enough to live in LDS, so the symbol search
// is one LDS look-up plus a short LDS scan; without it every symbol costs a binary search of dependent global loads.
struct RansDecTab { const int* row_off; const unsigned short* lut; const unsigned short* cdf16; };

RANS_HD const unsigned* r_decode(const unsigned* ptr, const unsigned* last, const int* idx, int ch0, long long n, int gl,
                                 long long stride, RansTab t, RansDecTab d, int* out) {
  unsigned long long x = (unsigned long long)ptr[0] | ((unsigned long long)ptr[1] << 32);   // (callers check: >= 2 words)
  ++ptr;
  unsigned nw = r_next(ptr, last);                    // look-ahead word
  const long long gm = (1ll << gl) - 1;
  for (long long base = 0; base < n; base += RB) {
    int ci[RB], size[RB], off[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const long long j = base + u;
      ci[u] = (j < n) ? (idx ? idx[(j >> gl) * stride + (j & gm)] : ch0 + (int)(j & gm)) : -1;
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      size[u] = ci[u] >= 0 ? t.sizes[ci[u]] : 0;
      off[u] = ci[u] >= 0 ? t.offsets[ci[u]] : 0;
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      if (ci[u] < 0) continue;
      const int* c = t.cdf + (long long)ci[u] * t.stride;
      const int max_value = size[u] - 2;
      const unsigned cum = (unsigned)(x & ((1u << R_PREC) - 1));
      int lo;
      unsigned start, freq;
      if (d.lut) {
        const unsigned short* c16 = d.cdf16 + d.row_off[ci[u]];
        lo = d.lut[ci[u] * 256 + (cum >> 8)];
        const int last = size[u] - 2;                 // c16[size-1] is 2^16 stored as 0: never compare against it
        while (lo < last && (unsigned)c16[lo + 1] <= cum) ++lo;
        start = c16[lo];
        freq = ((unsigned)c16[lo + 1] - start) & 0xFFFFu;
      } else {
        int hi = size[u] - 1;                         // last s with c[s] <= cum  (c[0] = 0, c[size-1] = 2^16 > cum)
        lo = 0;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((unsigned)c[mid] <= cum) lo = mid; else hi = mid; }
        start = (unsigned)c[lo]; freq = (unsigned)(c[lo + 1] - c[lo]);
      }
      x = (unsigned long long)freq * (x >> R_PREC) + cum - start;
      if (x < R_L) { x = (x << 32) | nw; nw = r_next(ptr, last); }
      int value = lo;
      if (value == max_value) {
        unsigned val = r_get_bits(x, ptr, nw, last);
        int nb = (int)val;
        while (val == R_MAXB && nb < 64) { val = r_get_bits(x, ptr, nw, last); nb += (int)val; }   // (the encoder writes <= 8 digits:
        unsigned raw = 0;                                                                         //  more is a corrupt stream, bounded)
        for (int j = 0; j < nb; ++j) { const unsigned b = r_get_bits(x, ptr, nw, last); if (j < 8) raw |= b << (j * R_BYP); }
        value = (int)(raw >> 1);
        value = (raw & 1) ? -value - 1 : value + max_value;
      }
      out[((base + u) >> gl) * stride + ((base + u) & gm)] = value + off[u];
    }
  }
  return ptr;
}

// ------------------------------------------------------------------------------------------
// host: CDF quantisation + single-stream coder
// ------------------------------------------------------------------------------------------
extern "C" int pcc_pmf_to_quantized_cdf(const float* h_pmf, int32_t n, int32_t precision, int32_t* h_cdf) {
  PCC_REQUIRE(h_pmf && h_cdf && n >= 1 && precision >= 1 && precision <= 16, "pcc_pmf_to_quantized_cdf: bad arguments");
  std::vector<uint32_t> cdf((size_t)n + 1);
  cdf[0] = 0;
  for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)roundf(h_pmf[i] * (float)(1 << precision));
  uint32_t total = 0;
  for (auto v : cdf) total += v;
  PCC_REQUIRE(total != 0, "pcc_pmf_to_quantized_cdf: pmf sums to zero");
  for (auto& v : cdf) v = (uint32_t)((((uint64_t)1 << precision) * v) / total);
  for (size_t i = 1; i < cdf.size(); ++i) cdf[i] += cdf[i - 1];
  cdf.back() = 1u << precision;
  for (int i = 0; i < (int)cdf.size() - 1; ++i) {
    if (cdf[i] == cdf[i + 1]) {
      uint32_t best_freq = ~0u;
      int best = -1;
      for (int j = 0; j < (int)cdf.size() - 1; ++j) {
        const uint32_t f = cdf[j + 1] - cdf[j];
        if (f > 1 && f < best_freq) { best_freq = f; best = j; }
      }
      PCC_REQUIRE(best != -1, "pcc_pmf_to_quantized_cdf: cannot give every symbol a non-zero frequency");
      if (best < i) for (int j = best + 1; j <= i; ++j) cdf[j]--;
      else for (int j = i + 1; j <= best; ++j) cdf[j]++;
    }
  }
  for (size_t i = 0; i < cdf.size(); ++i) h_cdf[i] = (int32_t)cdf[i];
  return PCC_OK;
}

// compact decoder table blob:  int32 rows | int32 total | int32 row_off[rows+1] | u16 lut[rows*256] | u16 cdf16[total]
extern "C" int64_t pcc_rans_dec_table_bytes(int32_t rows, const int32_t* h_sizes) {
  int64_t total = 0;
  for (int r = 0; r < rows; ++r) total += h_sizes[r];
  const int64_t b = 8 + 4 * ((int64_t)rows + 1) + 2 * (int64_t)rows * 256 + 2 * total;
  return (b + 15) / 16 * 16;
}

extern "C" int pcc_rans_build_dec_table(const int32_t* h_cdf, int32_t rows, int32_t cdf_stride, const int32_t* h_sizes,
                                        void* h_blob) {
  PCC_REQUIRE(h_cdf && h_sizes && h_blob && rows >= 1, "pcc_rans_build_dec_table: bad arguments");
  memset(h_blob, 0, (size_t)pcc_rans_dec_table_bytes(rows, h_sizes));
  int* hd = (int*)h_blob;
  int* row_off = hd + 2;
  unsigned short* lut = (unsigned short*)(row_off + rows + 1);
  unsigned short* cdf16 = lut + (size_t)rows * 256;
  int total = 0;
  for (int r = 0; r < rows; ++r) {
    row_off[r] = total;
    const int32_t* c = h_cdf + (int64_t)r * cdf_stride;
    for (int v = 0; v < h_sizes[r]; ++v) cdf16[total + v] = (unsigned short)(c[v] & 0xFFFF);   // 2^16 -> 0
    int sidx = 0;
    for (int b = 0; b < 256; ++b) {
      while (sidx + 1 < h_sizes[r] - 1 && c[sidx + 1] <= b * 256) ++sidx;
      lut[r * 256 + b] = (unsigned short)sidx;
    }
    total += h_sizes[r];
  }
  row_off[rows] = total;
  hd[0] = rows;
  hd[1] = total;
  return PCC_OK;
}

// division-free encoder entries, one per (row, value): layout [rows][cdf_stride] of 16-byte RansEncSym
extern "C" int pcc_rans_build_enc_table(const int32_t* h_cdf, int32_t rows, int32_t cdf_stride, const int32_t* h_sizes,
                                        void* h_table) {
  PCC_REQUIRE(h_cdf && h_sizes && h_table && rows >= 1, "pcc_rans_build_enc_table: bad arguments");
  static_assert(sizeof(RansEncSym) == 16, "RansEncSym must be 16 bytes");
  RansEncSym* out = (RansEncSym*)h_table;
  memset(out, 0, sizeof(RansEncSym) * (size_t)rows * cdf_stride);
  for (int r = 0; r < rows; ++r) {
    const int32_t* c = h_cdf + (int64_t)r * cdf_stride;
    for (int v = 0; v + 1 < h_sizes[r]; ++v) {
      const uint32_t start = (uint32_t)c[v], freq = (uint32_t)(c[v + 1] - c[v]);
      PCC_REQUIRE(freq >= 1 && freq < (1u << R_PREC), "pcc_rans_build_enc_table: bad frequency in row %d", r);
      RansEncSym& e = out[(int64_t)r * cdf_stride + v];
      e.cmpl_freq = (unsigned short)((1u << R_PREC) - freq);
      if (freq < 2) {
        e.rcp_freq = ~0ull; e.rcp_shift = 0; e.bias = start + (1u << R_PREC) - 1;
      } else {
        uint32_t shift = 0;
        while (freq > (1u << shift)) ++shift;
        uint64_t x0 = freq - 1;
        const uint64_t x1 = 1ull << (shift + 31);
        const uint64_t t1 = x1 / freq;
        x0 += (x1 % freq) << 32;
        const uint64_t t0 = x0 / freq;
        e.rcp_freq = t0 + (t1 << 32); e.rcp_shift = (unsigned short)(shift - 1); e.bias = start;
      }
    }
  }
  return PCC_OK;
}

extern "C" int64_t pcc_rans_max_bytes(int64_t n) { return (2 * n + 4) * 4; }

extern "C" int pcc_rans_encode_host(const int32_t* h_sym, const int32_t* h_idx, int64_t n, const int32_t* h_cdf,
                                    int32_t cdf_stride, const int32_t* h_sizes, const int32_t* h_offsets, uint8_t* h_out,
                                    int64_t cap, int64_t* h_nbytes) {
  PCC_REQUIRE(h_nbytes && h_out && h_cdf && h_sizes && h_offsets && (n == 0 || (h_sym && h_idx)), "pcc_rans_encode_host: NULL");
  PCC_REQUIRE(cap >= pcc_rans_max_bytes(n), "pcc_rans_encode_host: output capacity below pcc_rans_max_bytes");
  std::vector<unsigned> buf((size_t)(2 * n + 4));
  RansTab t{h_cdf, cdf_stride, h_sizes, h_offsets};
  unsigned* end = buf.data() + buf.size();
  unsigned* p = r_encode(h_sym, h_idx, 0, n, 0, 1, t, nullptr, end);
  *h_nbytes = (int64_t)(end - p) * 4;
  memcpy(h_out, p, (size_t)*h_nbytes);
  return PCC_OK;
}

extern "C" int pcc_rans_decode_host(const uint8_t* h_data, int64_t nbytes, const int32_t* h_idx, int64_t n,
                                    const int32_t* h_cdf, int32_t cdf_stride, const int32_t* h_sizes,
                                    const int32_t* h_offsets, int32_t* h_sym) {
  PCC_REQUIRE(h_data && h_cdf && h_sizes && h_offsets && (n == 0 || (h_sym && h_idx)), "pcc_rans_decode_host: NULL");
  PCC_REQUIRE(nbytes >= 8 && nbytes % 4 == 0, "pcc_rans_decode_host: truncated stream");
  const size_t words = (size_t)nbytes / 4;
  std::vector<unsigned> buf(words + 1, 0u);                  // + the look-ahead word of a stream consumed to its end
  memcpy(buf.data(), h_data, (size_t)nbytes);
  RansTab t{h_cdf, cdf_stride, h_sizes, h_offsets};
  const unsigned* p = r_decode(buf.data(), buf.data() + words, h_idx, 0, n, 0, 1, t, RansDecTab{nullptr, nullptr, nullptr}, h_sym);
  PCC_REQUIRE(p <= buf.data() + words, "pcc_rans_decode_host: corrupt stream (it ends before its symbols do)");
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// GPU: many independent streams, one per lane
//   The [n, channels] symbol matrix is cut into n_groups channel groups (2^gl adjacent channels each) times n_segments
//   row segments of R = ceil(n / n_segments) rows; stream s = segment * n_groups + group codes its tile row by row.
//   container: u32 n_streams | u32 nwords[n_streams] | words of stream 0 | words of stream 1 | ...
//   A lone wave issues about one instruction every 4-5 cycles and the coder state is a serial chain, so speed comes
//   from (a) many streams (segments), and (b) few instructions per symbol: 32-bit addressing, everything that does
//   not depend on the state resolved up front, tables in LDS addressed as LDS.
// ------------------------------------------------------------------------------------------
static constexpr int GB = 8;                        // symbols per prefetch batch in the GPU coders
static constexpr unsigned short R_ESC = 0x8000;     // RansEncSym::rcp_shift flag: bypass payload in pre_r

struct StreamGeom { int n, channels, n_groups, gl, R, n_streams; };

__device__ inline unsigned wave_incl_scan(unsigned v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ inline int wave_min(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
  return v;
}

// Pass 1 (one thread per symbol, fully parallel): resolve every symbol to its division-free encoder entry and its
// bypass payload, laid out [position in stream][stream] so that the sequential coder's reads are lane-contiguous.
__global__ void __launch_bounds__(256) k_rans_prepare(const int* __restrict__ sym, const int* __restrict__ idx, StreamGeom g,
                                                      RansTab t, const RansEncSym* __restrict__ enc,
                                                      RansEncSym* __restrict__ pre_e, unsigned* __restrict__ pre_r) {
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  if (e >= (unsigned)g.n * (unsigned)g.channels) return;
  const int i = (int)(e / (unsigned)g.channels);
  const int ch = (int)(e - (unsigned)i * (unsigned)g.channels);
  const int ci = idx ? idx[e] : ch;
  const int max_value = t.sizes[ci] - 2;
  int v = sym[e] - t.offsets[ci];
  unsigned raw = 0;
  bool esc = false;
  if (v < 0) { raw = (unsigned)(-2 * v - 1); v = max_value; esc = true; }
  else if (v >= max_value) { raw = (unsigned)(2 * (v - max_value)); v = max_value; esc = true; }
  const int seg = i / g.R;
  const int j = ((i - seg * g.R) << g.gl) + (ch & ((1 << g.gl) - 1));
  const size_t o = (size_t)j * g.n_streams + (seg * g.n_groups + (ch >> g.gl));
  RansEncSym es = enc[(size_t)ci * t.stride + v];
  if (esc) { es.rcp_shift |= R_ESC; pre_r[o] = raw; }
  pre_e[o] = es;
}

__device__ __forceinline__ void enc_one(unsigned long long& x, unsigned*& ptr, const RansEncSym& e,
                                        const unsigned* __restrict__ pre_r, size_t o) {
  if (e.rcp_shift & R_ESC) r_put_bypass(x, ptr, pre_r[o]);
  const unsigned freq = (1u << R_PREC) - e.cmpl_freq;
  if ((unsigned)(x >> 32) >= (freq << (31 - R_PREC))) { *--ptr = (unsigned)x; x >>= 32; }   // x >= freq << 47
  const unsigned long long q = __umul64hi(x, e.rcp_freq) >> (e.rcp_shift & 63);
  x = x + e.bias + q * e.cmpl_freq;
}

// Pass 2 (one lane per stream): the state recurrence, fed from the prepared entries in reverse symbol order.
// Batches of GB entries are loaded one batch ahead of their use.
__global__ void __launch_bounds__(64) k_rans_encode(const RansEncSym* __restrict__ pre_e, const unsigned* __restrict__ pre_r,
                                                    StreamGeom g, unsigned* __restrict__ scratch, long long cap_words,
                                                    int* __restrict__ nwords) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  const bool valid = s < g.n_streams;
  const int seg = s / g.n_groups;
  int rows = g.n - seg * g.R;
  rows = rows < 0 ? 0 : (rows > g.R ? g.R : rows);
  const int cnt = valid ? (rows << g.gl) : 0;
  const int cfast = __builtin_amdgcn_readfirstlane(wave_min(valid ? cnt : 0x7FFFFFFF)) / GB * GB;
  if (!valid) return;
  const int cmax = g.R << g.gl;
  unsigned* const end = scratch + (size_t)(s + 1) * cap_words;
  unsigned* ptr = end;
  unsigned long long x = R_L;
  const unsigned ns = (unsigned)g.n_streams;
  // ragged top (only where a wave spans the short last segment): per-lane predicate
  for (int j = cmax - 1; j >= cfast; --j) {
    if (j < cnt) {
      const size_t o = (size_t)j * ns + s;
      const RansEncSym e = pre_e[o];
      enc_one(x, ptr, e, pre_r, o);
    }
  }
  // full batches, wave-uniform trip count; two batches per trip with the buffers swapping roles (no register copy that
  // would make the coder wait for the entries just requested)
  auto fetch = [&](int base, RansEncSym (&e)[GB]) {      // entries base-1 ... base-GB (clamped at the stream start)
#pragma unroll
    for (int u = 0; u < GB; ++u) e[u] = pre_e[(size_t)max(base - 1 - u, 0) * ns + s];
  };
  auto run = [&](int base, const RansEncSym (&e)[GB]) {
#pragma unroll
    for (int u = 0; u < GB; ++u) enc_one(x, ptr, e[u], pre_r, (size_t)(base - 1 - u) * ns + s);
  };
  if (cfast > 0) {
    RansEncSym ea[GB], eb[GB];
    int base = cfast;
    fetch(base, ea);
    for (; base >= 2 * GB; base -= 2 * GB) {
      fetch(base - GB, eb);
      __builtin_amdgcn_sched_barrier(0);
      run(base, ea);
      __builtin_amdgcn_sched_barrier(0);
      fetch(base - 2 * GB, ea);
      __builtin_amdgcn_sched_barrier(0);
      run(base - GB, eb);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (base > 0) run(base, ea);                          // odd number of batches
  }
  ptr -= 2;
  ptr[0] = (unsigned)x;
  ptr[1] = (unsigned)(x >> 32);
  nwords[s] = (int)(end - ptr);
}

// one block per stream: position = sum of the lengths before it
__global__ void __launch_bounds__(64) k_rans_pack(const unsigned* __restrict__ scratch, long long cap_words,
                                                  const int* __restrict__ nwords, int n_streams,
                                                  unsigned* __restrict__ out, long long* __restrict__ d_nbytes) {
  const int s = blockIdx.x;
  unsigned part = 0;
  for (int i = threadIdx.x; i < s; i += 64) part += (unsigned)nwords[i];
  const unsigned off = __shfl(wave_incl_scan(part), 63, 64);
  const int len = nwords[s];
  if (threadIdx.x == 0) {
    out[1 + s] = (unsigned)len;
    if (s == 0) out[0] = (unsigned)n_streams;
    if (s == n_streams - 1) *d_nbytes = 4ll * (1 + n_streams + (long long)off + len);
  }
  const unsigned* src = scratch + (size_t)(s + 1) * cap_words - len;
  unsigned* dst = out + 1 + n_streams + off;
  for (int i = threadIdx.x; i < len; i += 64) dst[i] = src[i];
}

// ---- decoder -------------------------------------------------------------------------------
struct DecLds { const int* row_off; const unsigned short* lut; const unsigned short* c16; const int* offs; };

__device__ __forceinline__ unsigned dec_bits(unsigned long long& x, unsigned& nw, unsigned& pos,
                                             const unsigned* __restrict__ data, unsigned lim) {
  const unsigned v = (unsigned)x & R_MAXB;
  x >>= R_BYP;
  if (x < R_L) { x = (x << 32) | nw; ++pos; nw = data[min(pos, lim)]; }
  return v;
}

// rare path, kept out of line and register-to-register (by value) so the hot loop's state never touches memory
struct DecState { unsigned long long x; unsigned nw, pos; int value; };
__device__ __noinline__ DecState dec_bypass(unsigned long long x, unsigned nw, unsigned pos, const unsigned* __restrict__ data,
                                            unsigned lim, int max_value) {
  unsigned val = dec_bits(x, nw, pos, data, lim);
  int nb = (int)val;
  while (val == R_MAXB && nb < 64) { val = dec_bits(x, nw, pos, data, lim); nb += (int)val; }
  unsigned raw = 0;
  for (int j = 0; j < nb; ++j) { const unsigned b = dec_bits(x, nw, pos, data, lim); if (j < 8) raw |= b << (j * R_BYP); }
  const int value = (int)(raw >> 1);
  return DecState{x, nw, pos, (raw & 1) ? -value - 1 : value + max_value};
}

// One symbol: bucket look-up + short scan in the LDS tables, state update, renormalisation from the look-ahead word.
__device__ __forceinline__ int dec_one(int ci, unsigned long long& x, unsigned& nw, unsigned& pos,
                                       const unsigned* __restrict__ data, unsigned lim, const DecLds& d) {
  const int ro = d.row_off[ci];
  const int last = d.row_off[ci + 1] - ro - 2;          // c[last + 1] is 2^16 stored as 0: never compared against
  const unsigned cum = (unsigned)x & 0xFFFFu;
  int lo = d.lut[ci * 256 + (cum >> 8)];
  const unsigned short* c = d.c16 + ro;
  unsigned cur = c[lo], nxt = c[lo + 1];
  while (lo < last && nxt <= cum) { ++lo; cur = nxt; nxt = c[lo + 1]; }
  const unsigned freq = ((nxt - cur - 1u) & 0xFFFFu) + 1u;
  x = (unsigned long long)freq * (x >> R_PREC) + (cum - cur);
  if (x < R_L) { x = (x << 32) | nw; ++pos; nw = data[min(pos, lim)]; }
  int value = lo;
  if (lo == last) {
    const DecState b = dec_bypass(x, nw, pos, data, lim, last);
    x = b.x; nw = b.nw; pos = b.pos; value = b.value;
  }
  return value + d.offs[ci];
}

template <bool HAS_IDX>
__global__ void __launch_bounds__(256) k_rans_decode_lds(const unsigned* __restrict__ data, unsigned nwords_total,
                                                         const int* __restrict__ idx, StreamGeom g,
                                                         const int* __restrict__ offsets, const int* __restrict__ dec_blob,
                                                         int dec_words, int* __restrict__ out, int* __restrict__ status) {
  extern __shared__ int blob_s[];                      // decoder table | per-row value offsets | scan scratch
  const int tid = threadIdx.x;
  for (int i = tid; i < dec_words; i += 256) blob_s[i] = dec_blob[i];
  __syncthreads();
  const int rows_tab = blob_s[0];
  int* offs_s = blob_s + dec_words;
  for (int i = tid; i < rows_tab; i += 256) offs_s[i] = offsets[i];
  unsigned* red = (unsigned*)(offs_s + rows_tab);      // [0..3] per-wave totals of `part`, [4..7] of `len`
  DecLds d;
  d.row_off = blob_s + 2;
  d.lut = (const unsigned short*)(blob_s + 2 + rows_tab + 1);
  d.c16 = d.lut + rows_tab * 256;
  d.offs = offs_s;
  if ((int)data[0] != g.n_streams) { if (tid == 0) *status = 1; return; }
  // stream position: lengths of all earlier blocks (strided partial sums) + exclusive scan inside the block
  const int s = blockIdx.x * 256 + tid;
  unsigned part = 0;
  for (int i = tid; i < blockIdx.x * 256; i += 256) part += data[1 + i];
  const unsigned len = s < g.n_streams ? data[1 + s] : 0u;
  const unsigned pscan = wave_incl_scan(part), lscan = wave_incl_scan(len);
  if ((tid & 63) == 63) { red[tid >> 6] = pscan; red[4 + (tid >> 6)] = lscan; }
  __syncthreads();
  unsigned off = 1u + (unsigned)g.n_streams + red[0] + red[1] + red[2] + red[3] + (lscan - len);
  for (int w = 0; w < (tid >> 6); ++w) off += red[4 + w];
  const bool valid = s < g.n_streams;
  const int seg = s / g.n_groups, grp = s - seg * g.n_groups;
  int rows = g.n - seg * g.R;
  rows = rows < 0 ? 0 : (rows > g.R ? g.R : rows);
  const int cnt = valid ? (rows << g.gl) : 0;
  const int cfast = __builtin_amdgcn_readfirstlane(wave_min(valid ? cnt : 0x7FFFFFFF)) / GB * GB;
  if (!valid) return;
  if ((unsigned long long)off + len > nwords_total || len < 2) { *status = 2; return; }
  const unsigned lim = nwords_total + 1;               // the buffer is padded by two words
  const int gm = (1 << g.gl) - 1;
  const int ch0 = grp << g.gl;
  const unsigned lane_base = (unsigned)(seg * g.R) * (unsigned)g.channels + (unsigned)ch0;
  unsigned pos = off + 2;
  unsigned long long x = (unsigned long long)data[off] | ((unsigned long long)data[off + 1] << 32);
  unsigned nw = data[min(pos, lim)];
  int j = 0;
  // wave-uniform trip count, no per-symbol predicates.  The table rows of the NEXT batch are fetched while this batch is
  // decoded (two batches per trip, the buffers swapping roles: a copy would tie the decode to the loads just issued).
  auto addr = [&](int jj) { return lane_base + (unsigned)(jj >> g.gl) * (unsigned)g.channels + (unsigned)(jj & gm); };
  auto fetch = [&](int j0, int (&ci)[GB]) {
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      const int jj = min(j0 + u, cfast - 1);            // past the end: re-read the last row (unused)
      ci[u] = HAS_IDX ? idx[addr(jj)] : ch0 + (jj & gm);
    }
  };
  auto run = [&](int j0, const int (&ci)[GB]) {
#pragma unroll
    for (int u = 0; u < GB; ++u) out[addr(j0 + u)] = dec_one(ci[u], x, nw, pos, data, lim, d);
  };
  if (cfast > 0) {
    int ca[GB], cb[GB];
    fetch(0, ca);
    for (; j + 2 * GB <= cfast; j += 2 * GB) {
      fetch(j + GB, cb);
      __builtin_amdgcn_sched_barrier(0);
      run(j, ca);
      __builtin_amdgcn_sched_barrier(0);
      fetch(j + 2 * GB, ca);
      __builtin_amdgcn_sched_barrier(0);
      run(j + GB, cb);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (j < cfast) { run(j, ca); j += GB; }              // odd number of batches
  }
  const int cmax = g.R << g.gl;
  for (; j < cmax; ++j) {
    if (j < cnt) {
      const unsigned a = lane_base + (unsigned)(j >> g.gl) * (unsigned)g.channels + (unsigned)(j & gm);
      const int ci = HAS_IDX ? idx[a] : ch0 + (j & gm);
      out[a] = dec_one(ci, x, nw, pos, data, lim, d);
    }
  }
  if (pos - 1 - off > len) *status = 3;                // consumed past its own stream: corrupt input
}

// fallback when the decoder table does not fit LDS: generic search in global memory
__global__ void __launch_bounds__(64) k_rans_decode_global(const unsigned* __restrict__ data, long long nwords_total,
                                                           const int* __restrict__ idx, StreamGeom g, RansTab t,
                                                           int* __restrict__ out, int* __restrict__ status) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= g.n_streams) return;
  if ((int)data[0] != g.n_streams) { *status = 1; return; }
  long long off = 1 + g.n_streams;
  for (int i = 0; i < s; ++i) off += data[1 + i];
  const long long len = data[1 + s];
  if (off + len > nwords_total || len < 2) { *status = 2; return; }
  const int seg = s / g.n_groups, grp = s - seg * g.n_groups;
  long long rows = (long long)g.n - (long long)seg * g.R;
  rows = rows < 0 ? 0 : (rows > g.R ? g.R : rows);
  const long long e0 = (long long)seg * g.R * g.channels + ((long long)grp << g.gl);
  const unsigned* p = r_decode(data + off, data + nwords_total - 1, idx ? idx + e0 : nullptr, grp << g.gl, rows << g.gl, g.gl,
                               g.channels, t, RansDecTab{nullptr, nullptr, nullptr}, out + e0);
  if (p - (data + off) > len) *status = 3;
}

// n = symbols per stream (the longest one)
extern "C" int64_t pcc_rans_container_max_bytes(int64_t n, int32_t n_streams) {
  return 4 * (1 + (int64_t)n_streams) + (int64_t)n_streams * pcc_rans_max_bytes(n);
}

extern "C" size_t pcc_rans_streams_ws_bytes(int64_t n, int32_t n_streams) {
  // word scratch per stream + stream lengths + prepared entries (16 B) and bypass payloads (4 B) per symbol
  return (size_t)n_streams * (size_t)(2 * n + 4) * 4 + pcc_align_up((size_t)n_streams * 4) +
         pcc_align_up((size_t)n_streams * (size_t)n * 16) + pcc_align_up((size_t)n_streams * (size_t)n * 4) + 1024;
}

static int stream_geom(int64_t n, int channels, int n_groups, int n_segments, StreamGeom* g) {
  if (n_groups < 1 || n_segments < 1 || channels < 1 || channels % n_groups) return -1;
  const int grp = channels / n_groups;
  int l = 0;
  while ((1 << l) < grp) ++l;
  if ((1 << l) != grp) return -1;
  if ((int64_t)n_groups * n_segments > 65536 || n * channels >= (1ll << 31)) return -1;
  g->n = (int)n; g->channels = channels; g->n_groups = n_groups; g->gl = l;
  g->R = (int)((n + n_segments - 1) / n_segments);
  if (g->R < 1) g->R = 1;
  g->n_streams = n_groups * n_segments;
  return 0;
}

extern "C" int64_t pcc_rans_stream_symbols(int64_t n, int32_t channels, int32_t n_groups, int32_t n_segments) {
  StreamGeom g;
  if (stream_geom(n, channels, n_groups, n_segments, &g)) return -1;
  return (int64_t)g.R << g.gl;
}

// Payload estimate of a symbol matrix under the given tables, before any stream geometry is chosen: sum over the symbols
// of round(256 * (16 - log2 freq)) (+ 256 * 8 per 4-bit bypass digit pair for escapes), an integer, so the total does not
// depend on the order of the (integer) atomic adds -- the encoder picks its stream count from it (framing <= 2 % of the
// payload) and must pick the same count every time it sees the same symbols.
__global__ void __launch_bounds__(256) k_rans_estimate(const int* __restrict__ sym, const int* __restrict__ idx, long long total,
                                                       int channels, RansTab t, unsigned long long* __restrict__ bits256) {
  // grid-stride: one 64-bit atomic per workgroup at the end, and few workgroups (7 450 same-address atomics of the
  // element-per-thread form cost ~75 of the kernel's 92 us; the sum is an integer, so its order does not matter)
  unsigned long long acc = 0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    unsigned b = 0;
    const int ci = idx ? idx[e] : (int)(e % channels);
    const int max_value = t.sizes[ci] - 2;
    int v = sym[e] - t.offsets[ci];
    unsigned raw = 0;
    if (v < 0) { raw = (unsigned)(-2 * v - 1); v = max_value; }
    else if (v >= max_value) { raw = (unsigned)(2 * (v - max_value)); v = max_value; }
    const int* row = t.cdf + (size_t)ci * t.stride;
    const int freq = max(row[v + 1] - row[v], 1);
    b = (unsigned)__float2int_rn(256.f * (16.f - __log2f((float)freq)));
    if (v == max_value) b += 256u * 4u * (unsigned)((32 - __clz((int)(raw | 1u)) + 3) / 4 + 1);   // bypass digits, 4 bits each
    acc += b;
  }
  unsigned long long w = acc;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) w += __shfl_xor(w, d, 64);
  __shared__ unsigned long long part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(bits256, part[0] + part[1] + part[2] + part[3]);
}

extern "C" int pcc_rans_estimate_bits(const int32_t* sym, const int32_t* idx, int64_t n, int32_t channels, const int32_t* cdf,
                                      int32_t cdf_stride, const int32_t* sizes, const int32_t* offsets, int64_t* d_bits256,
                                      void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(sym && cdf && sizes && offsets && d_bits256 && channels >= 1 && n >= 0, "pcc_rans_estimate_bits: bad arguments");
  PCC_CHECK_HIP(hipMemsetAsync(d_bits256, 0, 8, s));
  if (n == 0) return PCC_OK;
  RansTab t{cdf, cdf_stride, sizes, offsets};
  const long long total = (long long)n * channels;
  const long long blocks = pcc_cdiv(total, 256);
  k_rans_estimate<<<(unsigned)(blocks < 1024 ? blocks : 1024), 256, 0, s>>>(sym, idx, total, channels, t, (unsigned long long*)d_bits256);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_rans_encode_streams(const int32_t* sym, const int32_t* idx, int64_t n, int32_t channels,
                                       int32_t n_groups, int32_t n_segments, const int32_t* cdf,
                                       int32_t cdf_stride, const int32_t* sizes, const int32_t* offsets,
                                       const void* enc_table, uint8_t* out, int64_t* d_nbytes, void* ws,
                                       size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(sym && cdf && sizes && offsets && out && d_nbytes && ws, "pcc_rans_encode_streams: NULL array");
  PCC_REQUIRE(enc_table, "pcc_rans_encode_streams: enc_table is required (pcc_rans_build_enc_table)");
  StreamGeom g;
  PCC_REQUIRE(n >= 0 && stream_geom(n, channels, n_groups, n_segments, &g) == 0,
              "pcc_rans_encode_streams: bad geometry: %d channels, %d groups (power-of-two size), %d segments, "
              "at most 65536 streams and 2^31 symbols", channels, n_groups, n_segments);
  const int64_t per_stream = (int64_t)g.R << g.gl;
  if (ws_bytes < pcc_rans_streams_ws_bytes(per_stream, g.n_streams)) {
    pcc_set_error("pcc_rans_encode_streams: workspace too small");
    return PCC_EWS;
  }
  const long long cap = 2 * per_stream + 4;
  char* p = (char*)ws;
  unsigned* scratch = (unsigned*)p;     p += (size_t)g.n_streams * cap * 4;
  int* nwords = (int*)p;                p += pcc_align_up((size_t)g.n_streams * 4);
  RansEncSym* pre_e = (RansEncSym*)p;   p += pcc_align_up((size_t)g.n_streams * (size_t)per_stream * 16);
  unsigned* pre_r = (unsigned*)p;
  RansTab t{cdf, cdf_stride, sizes, offsets};
  if (n > 0) {
    k_rans_prepare<<<(unsigned)pcc_cdiv(n * channels, 256), 256, 0, s>>>(sym, idx, g, t, (const RansEncSym*)enc_table,
                                                                        pre_e, pre_r);
    PCC_LAUNCH_CHECK();
  }
  k_rans_encode<<<(unsigned)pcc_cdiv(g.n_streams, 64), 64, 0, s>>>(pre_e, pre_r, g, scratch, cap, nwords);
  PCC_LAUNCH_CHECK();
  k_rans_pack<<<g.n_streams, 64, 0, s>>>(scratch, cap, nwords, g.n_streams, (unsigned*)out, (long long*)d_nbytes);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_rans_decode_streams(const uint8_t* data, int64_t nbytes, const int32_t* idx, int64_t n,
                                       int32_t channels, int32_t n_groups, int32_t n_segments,
                                       const int32_t* cdf, int32_t cdf_stride, const int32_t* sizes,
                                       const int32_t* offsets, const void* dec_table, int64_t dec_bytes, int32_t* sym_out,
                                       int32_t* d_status, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(data && cdf && sizes && offsets && sym_out && d_status, "pcc_rans_decode_streams: NULL array");
  StreamGeom g;
  PCC_REQUIRE(n >= 0 && stream_geom(n, channels, n_groups, n_segments, &g) == 0,
              "pcc_rans_decode_streams: bad geometry: %d channels, %d groups (power-of-two size), %d segments, "
              "at most 65536 streams and 2^31 symbols", channels, n_groups, n_segments);
  PCC_REQUIRE(nbytes >= 4 * (1 + (int64_t)g.n_streams) && nbytes % 4 == 0 && nbytes < (1ll << 33),
              "pcc_rans_decode_streams: truncated container");
  PCC_CHECK_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t), s));
  RansTab t{cdf, cdf_stride, sizes, offsets};
  // decoder table in LDS when it fits (Gaussian scale table: ~90 KB; factorised prior: ~110 KB)
  const bool in_lds = dec_table && dec_bytes > 0 && dec_bytes <= 150 * 1024 && dec_bytes % 4 == 0;
  if (in_lds) {
    static unsigned long long attr_set = 0;                      // one bit per device (hipFuncSetAttribute is per device)
    int dev = 0;
    PCC_CHECK_HIP(hipGetDevice(&dev));
    if (!(attr_set >> (dev & 63) & 1ull)) {
      PCC_CHECK_HIP(hipFuncSetAttribute((const void*)k_rans_decode_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      PCC_CHECK_HIP(hipFuncSetAttribute((const void*)k_rans_decode_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set |= 1ull << (dev & 63);
    }
    // LDS: table + per-row offsets (row count = first word of the table, bounded by its size) + scan scratch
    const size_t lds = (size_t)dec_bytes + (size_t)dec_bytes / 128 + 64;
    const unsigned blocks = (unsigned)pcc_cdiv(g.n_streams, 256);
    if (idx)
      k_rans_decode_lds<true><<<blocks, 256, lds, s>>>((const unsigned*)data, (unsigned)(nbytes / 4), idx, g, offsets,
                                                      (const int*)dec_table, (int)(dec_bytes / 4), sym_out, d_status);
    else
      k_rans_decode_lds<false><<<blocks, 256, lds, s>>>((const unsigned*)data, (unsigned)(nbytes / 4), idx, g, offsets,
                                                       (const int*)dec_table, (int)(dec_bytes / 4), sym_out, d_status);
  } else {
    k_rans_decode_global<<<(unsigned)pcc_cdiv(g.n_streams, 64), 64, 0, s>>>((const unsigned*)data, nbytes / 4, idx, g, t,
                                                                           sym_out, d_status);
  }
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}
