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
RANS_HD unsigned r_get_bits(unsigned long long& x, const unsigned*& ptr, unsigned& nw) {
  const unsigned v = (unsigned)(x & R_MAXB);
  x >>= R_BYP;
  if (x < R_L) { x = (x << 32) | nw; nw = *++ptr; }
  return v;
}

// decodes n symbols forward; returns the pointer past the last word consumed.  `d` (nullable pointers) is the
// compact decoder table: 16-bit CDF rows back to back (row_off), plus per row a 256-entry bucket table
// lut[row*256 + (cum >> 8)] = last s with cdf[s] <= (cum & ~255).  Small enough to live in LDS, so the symbol search
// is one LDS look-up plus a short LDS scan; without it every symbol costs a binary search of dependent global loads.
struct RansDecTab { const int* row_off; const unsigned short* lut; const unsigned short* cdf16; };

RANS_HD const unsigned* r_decode(const unsigned* ptr, const int* idx, int ch0, long long n, int gl, long long stride,
                                 RansTab t, RansDecTab d, int* out) {
  unsigned long long x = (unsigned long long)ptr[0] | ((unsigned long long)ptr[1] << 32);
  ptr += 2;
  unsigned nw = *ptr;                                 // look-ahead word (the buffer is padded by one word)
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
      if (x < R_L) { x = (x << 32) | nw; nw = *++ptr; }
      int value = lo;
      if (value == max_value) {
        unsigned val = r_get_bits(x, ptr, nw);
        int nb = (int)val;
        while (val == R_MAXB) { val = r_get_bits(x, ptr, nw); nb += (int)val; }
        unsigned raw = 0;
        for (int j = 0; j < nb; ++j) raw |= r_get_bits(x, ptr, nw) << (j * R_BYP);
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
  std::vector<unsigned> buf((size_t)nbytes / 4 + 4, 0u);     // zero padding: a corrupt stream cannot read out of bounds far
  memcpy(buf.data(), h_data, (size_t)nbytes);
  RansTab t{h_cdf, cdf_stride, h_sizes, h_offsets};
  r_decode(buf.data(), h_idx, 0, n, 0, 1, t, RansDecTab{nullptr, nullptr, nullptr}, h_sym);
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// GPU: one stream per channel
//   container: u32 n_streams | u32 nwords[n_streams] | words of stream 0 | words of stream 1 | ...
// ------------------------------------------------------------------------------------------
// Pass 1 (one thread per symbol, fully parallel): resolve every symbol to its division-free encoder entry and its
// bypass payload, laid out [position in stream][stream] so that the sequential coder's reads are lane-contiguous.
__global__ void __launch_bounds__(256) k_rans_prepare(const int* __restrict__ sym, const int* __restrict__ idx, long long n,
                                                      int channels, int n_streams, int gl, RansTab t,
                                                      const RansEncSym* __restrict__ enc, RansEncSym* __restrict__ pre_e,
                                                      unsigned* __restrict__ pre_r) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * channels) return;
  const long long i = e / channels;
  const int ch = (int)(e - i * channels);
  const int ci = idx ? idx[e] : ch;
  const int max_value = t.sizes[ci] - 2;
  int v = sym[e] - t.offsets[ci];
  unsigned raw = 0xFFFFFFFFu;
  if (v < 0) { raw = (unsigned)(-2 * v - 1); v = max_value; }
  else if (v >= max_value) { raw = (unsigned)(2 * (v - max_value)); v = max_value; }
  const long long j = (i << gl) + (ch & ((1 << gl) - 1));
  const long long o = j * n_streams + (ch >> gl);
  pre_e[o] = enc[(long long)ci * t.stride + v];
  pre_r[o] = raw;
}

// Pass 2 (one lane per stream): the state recurrence, fed from the prepared entries.  Batches of RB entries are
// loaded one batch ahead of their use, so the only serial dependence left is the coder state itself.
__global__ void __launch_bounds__(64) k_rans_encode(const RansEncSym* __restrict__ pre_e, const unsigned* __restrict__ pre_r,
                                                    long long n_per_stream, int n_streams,
                                                    unsigned* __restrict__ scratch, long long cap_words,
                                                    int* __restrict__ nwords) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= n_streams) return;
  unsigned* end = scratch + (long long)(s + 1) * cap_words;
  unsigned* ptr = end;
  unsigned long long x = R_L;
  RansEncSym ea[RB], eb[RB];
  unsigned ra[RB], rb[RB];
  auto load = [&](long long base, RansEncSym (&e)[RB], unsigned (&r)[RB]) {
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const long long j = base - 1 - u;
      r[u] = 0xFFFFFFFEu;                                     // marker: past the start of the stream
      if (j >= 0) { e[u] = pre_e[j * n_streams + s]; r[u] = pre_r[j * n_streams + s]; }
    }
  };
  load(n_per_stream, ea, ra);
  for (long long base = n_per_stream; base > 0; base -= RB) {
    load(base - RB, eb, rb);                                  // next batch in flight while this one is coded
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      if (ra[u] != 0xFFFFFFFEu) {
        if (ra[u] != 0xFFFFFFFFu) r_put_bypass(x, ptr, ra[u]);
        r_put_sym(x, ptr, ea[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) { ea[u] = eb[u]; ra[u] = rb[u]; }
  }
  ptr -= 2;
  ptr[0] = (unsigned)x;
  ptr[1] = (unsigned)(x >> 32);
  nwords[s] = (int)(end - ptr);
}

__global__ void __launch_bounds__(256) k_rans_pack(const unsigned* __restrict__ scratch, long long cap_words,
                                                   const int* __restrict__ nwords, int n_streams,
                                                   unsigned* __restrict__ out, long long* __restrict__ d_nbytes) {
  __shared__ long long total;
  // few hundred streams: a serial prefix by one thread is cheaper than a scan launch
  extern __shared__ long long offs[];
  if (threadIdx.x == 0) {
    long long run = 0;
    for (int s = 0; s < n_streams; ++s) { offs[s] = run; run += nwords[s]; }
    total = run;
    out[0] = (unsigned)n_streams;
    *d_nbytes = 4ll * (1 + n_streams + run);
  }
  __syncthreads();
  for (int s = threadIdx.x; s < n_streams; s += 256) out[1 + s] = (unsigned)nwords[s];
  unsigned* dst = out + 1 + n_streams;
  for (int s = 0; s < n_streams; ++s) {
    const unsigned* src = scratch + (long long)(s + 1) * cap_words - nwords[s];
    for (int i = threadIdx.x; i < nwords[s]; i += 256) dst[offs[s] + i] = src[i];
  }
  (void)total;
}

__global__ void __launch_bounds__(64) k_rans_decode(const unsigned* __restrict__ data, long long nwords_total,
                                                    const int* __restrict__ idx, long long n, int n_streams,
                                                    int gl, long long row_stride, RansTab t,
                                                    const int* __restrict__ dec_blob, int dec_words_lds,
                                                    int* __restrict__ out, int* __restrict__ status) {
  extern __shared__ int blob_s[];
  RansDecTab d{nullptr, nullptr, nullptr};
  if (dec_blob) {
    const int* b = dec_blob;
    if (dec_words_lds > 0) {                            // the whole table fits LDS: symbol search at LDS latency
      for (int i = threadIdx.x; i < dec_words_lds; i += blockDim.x) blob_s[i] = dec_blob[i];
      __syncthreads();
      b = blob_s;
    }
    const int rows = b[0];
    d.row_off = b + 2;
    d.lut = (const unsigned short*)(b + 2 + rows + 1);
    d.cdf16 = d.lut + (long long)rows * 256;
  }
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_streams) return;
  if ((int)data[0] != n_streams) { *status = 1; return; }
  long long off = 1 + n_streams;
  for (int i = 0; i < s; ++i) off += data[1 + i];
  const long long len = data[1 + s];
  if (off + len > nwords_total || len < 2) { *status = 2; return; }
  const long long ch0 = (long long)s << gl;
  const unsigned* p = r_decode(data + off, idx ? idx + ch0 : nullptr, (int)ch0, n << gl, gl, row_stride, t, d, out + ch0);
  if (p - (data + off) > len) *status = 3;      // read past its own stream: corrupt input
}

// n = symbols per stream
extern "C" int64_t pcc_rans_container_max_bytes(int64_t n, int32_t n_streams) {
  return 4 * (1 + (int64_t)n_streams) + (int64_t)n_streams * pcc_rans_max_bytes(n);
}

extern "C" size_t pcc_rans_streams_ws_bytes(int64_t n, int32_t n_streams) {
  // word scratch per stream + stream lengths + prepared entries (16 B) and bypass payloads (4 B) per symbol
  return (size_t)n_streams * (size_t)(2 * n + 4) * 4 + pcc_align_up((size_t)n_streams * 4) +
         pcc_align_up((size_t)n_streams * (size_t)n * 16) + pcc_align_up((size_t)n_streams * (size_t)n * 4) + 1024;
}

static int group_log2(int channels, int n_streams) {
  if (n_streams < 1 || channels % n_streams) return -1;
  const int g = channels / n_streams;
  int l = 0;
  while ((1 << l) < g) ++l;
  return (1 << l) == g ? l : -1;
}

extern "C" int pcc_rans_encode_streams(const int32_t* sym, const int32_t* idx, int64_t n, int32_t channels,
                                       int32_t n_streams, const int32_t* cdf,
                                       int32_t cdf_stride, const int32_t* sizes, const int32_t* offsets,
                                       const void* enc_table, uint8_t* out, int64_t* d_nbytes, void* ws,
                                       size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(sym && cdf && sizes && offsets && out && d_nbytes && ws, "pcc_rans_encode_streams: NULL array");
  PCC_REQUIRE(enc_table, "pcc_rans_encode_streams: enc_table is required (pcc_rans_build_enc_table)");
  PCC_REQUIRE(n >= 0 && n_streams >= 1 && n_streams <= 4096, "pcc_rans_encode_streams: bad stream count %d", n_streams);
  const int gl = group_log2(channels, n_streams);
  PCC_REQUIRE(gl >= 0, "pcc_rans_encode_streams: %d channels do not split into %d power-of-two groups", channels, n_streams);
  const int64_t per_stream = n << gl;
  if (ws_bytes < pcc_rans_streams_ws_bytes(per_stream, n_streams)) {
    pcc_set_error("pcc_rans_encode_streams: workspace too small");
    return PCC_EWS;
  }
  const long long cap = 2 * per_stream + 4;
  char* p = (char*)ws;
  unsigned* scratch = (unsigned*)p;     p += (size_t)n_streams * cap * 4;
  int* nwords = (int*)p;                p += pcc_align_up((size_t)n_streams * 4);
  RansEncSym* pre_e = (RansEncSym*)p;   p += pcc_align_up((size_t)n_streams * (size_t)per_stream * 16);
  unsigned* pre_r = (unsigned*)p;
  RansTab t{cdf, cdf_stride, sizes, offsets};
  if (n > 0) {
    k_rans_prepare<<<(unsigned)pcc_cdiv(n * channels, 256), 256, 0, s>>>(sym, idx, n, channels, n_streams, gl, t,
                                                                        (const RansEncSym*)enc_table, pre_e, pre_r);
    PCC_LAUNCH_CHECK();
  }
  k_rans_encode<<<(unsigned)pcc_cdiv(n_streams, 64), 64, 0, s>>>(pre_e, pre_r, per_stream, n_streams, scratch, cap, nwords);
  PCC_LAUNCH_CHECK();
  k_rans_pack<<<1, 256, (size_t)n_streams * sizeof(long long), s>>>(scratch, cap, nwords, n_streams, (unsigned*)out, (long long*)d_nbytes);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_rans_decode_streams(const uint8_t* data, int64_t nbytes, const int32_t* idx, int64_t n,
                                       int32_t channels, int32_t n_streams,
                                       const int32_t* cdf, int32_t cdf_stride, const int32_t* sizes,
                                       const int32_t* offsets, const void* dec_table, int64_t dec_bytes, int32_t* sym_out,
                                       int32_t* d_status, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(data && cdf && sizes && offsets && sym_out && d_status, "pcc_rans_decode_streams: NULL array");
  PCC_REQUIRE(n >= 0 && n_streams >= 1 && n_streams <= 4096, "pcc_rans_decode_streams: bad stream count %d", n_streams);
  PCC_REQUIRE(nbytes >= 4 * (1 + (int64_t)n_streams) && nbytes % 4 == 0, "pcc_rans_decode_streams: truncated container");
  const int gl = group_log2(channels, n_streams);
  PCC_REQUIRE(gl >= 0, "pcc_rans_decode_streams: %d channels do not split into %d power-of-two groups", channels, n_streams);
  PCC_CHECK_HIP(hipMemsetAsync(d_status, 0, sizeof(int32_t), s));
  RansTab t{cdf, cdf_stride, sizes, offsets};
  // decoder table in LDS when it fits (Gaussian scale table: ~90 KB; factorised prior: ~110 KB)
  const bool in_lds = dec_table && dec_bytes > 0 && dec_bytes <= 150 * 1024;
  const size_t lds = in_lds ? (size_t)dec_bytes : 0;
  static bool attr_set = false;
  if (in_lds && !attr_set) {
    PCC_CHECK_HIP(hipFuncSetAttribute((const void*)k_rans_decode, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set = true;
  }
  k_rans_decode<<<(unsigned)pcc_cdiv(n_streams, 64), 64, lds, s>>>((const unsigned*)data, nbytes / 4, idx, n, n_streams,
                                                                    gl, channels, t, (const int*)dec_table,
                                                                    in_lds ? (int)(dec_bytes / 4) : 0,
                                                                    sym_out, d_status);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}
