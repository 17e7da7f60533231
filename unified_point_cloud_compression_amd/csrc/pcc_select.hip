// Top-k occupancy selection (radix select, no sort), row compaction (pruning) and exact-match
// row lookup / gather.  HBM-bound; ballots + scans, no global atomics on the data path except the
// 256-bin select histogram.
#include "pcc_common.h"

// monotone float -> uint32 (ascending); -0.0 is folded onto +0.0 so equal floats tie
__device__ inline unsigned f2u(float f) {
  if (f == 0.f) f = 0.f;
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct SelState { unsigned prefix, mask; int remaining; int pad; };

// Radix select in three passes of 11 + 11 + 10 bits (round 3; four passes of 8 before): a pass is one read of the logits with a
// histogram of the digit below the prefix found so far, then one workgroup picks the digit holding the `remaining`-th largest
// candidate.  With 11 bits the first digit is sign + exponent + 2 mantissa bits, so occupancy logits no longer pile onto two or
// three bins of the LDS histogram (same-address atomics of a wave serialise).
static constexpr int SEL_MAXBINS = 2048;
static constexpr int SEL_WS_HDR = 256 + SEL_MAXBINS * 4 + 256;      // state | histogram | pad
static constexpr int SEL_B = 2048;                                    // rows per workgroup of the count / compact passes (256 x 8)

__global__ void k_sel_init(SelState* st, int* hist, int k) {
  if (threadIdx.x == 0) { st->prefix = 0; st->mask = 0; st->remaining = k; st->pad = 0; }
  for (int i = threadIdx.x; i < SEL_MAXBINS; i += blockDim.x) hist[i] = 0;
}

template <int BITS>
__global__ void __launch_bounds__(256) k_sel_hist(const float* __restrict__ logits, long long stride, long long n,
                                                  int shift, const SelState* __restrict__ st, int* __restrict__ hist) {
  constexpr int NB = 1 << BITS;
  __shared__ int h[NB];
  for (int i = threadIdx.x; i < NB; i += 256) h[i] = 0;
  __syncthreads();
  const unsigned prefix = st->prefix, mask = st->mask;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const unsigned u = f2u(logits[i * stride]);
    if ((u & mask) == prefix) atomicAdd(&h[(u >> shift) & (NB - 1)], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NB; i += 256)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

// pick the digit holding the `remaining`-th largest candidate (bins walked from the top); clears hist for the next pass.
// 256 threads: thread t owns the NB/256 bins below NB - t * per; an exclusive scan over the threads finds the owner.
__global__ void __launch_bounds__(256) k_sel_pick(SelState* st, int* hist, int shift, int bits) {
  __shared__ int wsum[4];
  __shared__ int h[SEL_MAXBINS];
  const int nb = 1 << bits, per = nb >> 8;
  for (int i = threadIdx.x; i < nb; i += 256) { h[i] = hist[i]; hist[i] = 0; }
  __syncthreads();
  const int top = nb - 1 - (int)threadIdx.x * per;                    // this thread's bins: top, top-1, ..., top-per+1
  int mine = 0;
  for (int q = 0; q < per; ++q) mine += h[top - q];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int before = inc - mine;
  for (int q = 0; q < w; ++q) before += wsum[q];
  const int rem = st->remaining;                                       // (read by everyone before anyone writes: barrier below)
  const bool last = threadIdx.x == 255;
  __syncthreads();
  // the owner: cumulative count reaches `rem` inside its bins; if it is never reached (rem > population, cannot happen for
  // 0 < k < n) the last thread settles on bin 0, as the sequential walk did
  if ((before < rem && before + mine >= rem) || (last && before + mine < rem)) {
    int cum = before, d = top;
    for (int q = 0; q < per; ++q, --d) {
      if (cum + h[d] >= rem || d == 0) break;
      cum += h[d];
    }
    st->remaining = rem - cum;
    st->prefix |= (unsigned)d << shift;
    st->mask |= (unsigned)(nb - 1) << shift;
  }
}

// ---- mask + stable compaction in three launches (after the select: T = st->prefix is the k-th largest value) --------------
// keep[i] = u_i > T  or  (u_i == T and fewer than `remaining` equal values precede i): the rows of a workgroup need the number of
// equal values and of kept rows before it -- both follow from per-workgroup (greater, equal) counts by one small scan.
__device__ inline int sel_block_scan(int v, int* total, int* wsum) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int sv = wsum[q]; if (q < w) base += sv; tot += sv; }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void __launch_bounds__(256) k_sel_count(const float* __restrict__ logits, long long stride, long long n,
                                                   const SelState* __restrict__ st, int2* __restrict__ cnt) {
  __shared__ int wsum[4];
  const unsigned T = st->prefix;
  const long long base = (long long)blockIdx.x * SEL_B + (long long)threadIdx.x * 8;
  int gt = 0, eq = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (base + q < n) {
      const unsigned u = f2u(logits[(base + q) * stride]);
      gt += u > T; eq += u == T;
    }
  int tg, te;
  sel_block_scan(gt, &tg, wsum);
  sel_block_scan(eq, &te, wsum);
  if (threadIdx.x == 0) cnt[blockIdx.x] = make_int2(tg, te);
}

// one workgroup: per counted workgroup b the equal values before it and the rows kept before it (exclusive scans, chunked)
__global__ void __launch_bounds__(256) k_sel_offsets(const int2* __restrict__ cnt, int nblk, const SelState* __restrict__ st,
                                                     int2* __restrict__ before) {
  __shared__ int wsum[4];
  const int rem = st->remaining;
  int carry_eq = 0, carry_keep = 0;
  for (int b0 = 0; b0 < nblk; b0 += 256) {
    const int b = b0 + (int)threadIdx.x;
    const int2 c = b < nblk ? cnt[b] : make_int2(0, 0);
    int tot;
    const int eqb = carry_eq + sel_block_scan(c.y, &tot, wsum);
    carry_eq += tot;
    const int room = rem - eqb;
    const int kept = c.x + (room <= 0 ? 0 : (room < c.y ? room : c.y));
    const int keepb = carry_keep + sel_block_scan(kept, &tot, wsum);
    carry_keep += tot;
    if (b < nblk) before[b] = make_int2(eqb, keepb);
  }
}

__global__ void __launch_bounds__(256) k_sel_compact(const float* __restrict__ logits, long long stride, long long n,
                                                     const SelState* __restrict__ st, const int2* __restrict__ before,
                                                     const int64_t* __restrict__ keys, unsigned char* __restrict__ mask,
                                                     int64_t* __restrict__ keys_out) {
  __shared__ int wsum[4];
  const unsigned T = st->prefix;
  const int rem = st->remaining;
  const int2 bf = before[blockIdx.x];
  const long long base = (long long)blockIdx.x * SEL_B + (long long)threadIdx.x * 8;
  unsigned u[8];
  int eq = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    u[q] = base + q < n ? f2u(logits[(base + q) * stride]) : 0u;
    eq += (base + q < n) && u[q] == T;
  }
  int tot;
  int eq_rank = bf.x + sel_block_scan(eq, &tot, wsum);
  bool keep[8];
  int kc = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const bool in = base + q < n;
    const bool e = in && u[q] == T;
    keep[q] = in && (u[q] > T || (e && eq_rank < rem));
    eq_rank += e;
    kc += keep[q];
  }
  int pos = bf.y + sel_block_scan(kc, &tot, wsum);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (base + q >= n) break;
    mask[base + q] = keep[q] ? 1 : 0;
    if (keep[q]) {
      if (keys_out) keys_out[pos] = keys[base + q];
      ++pos;
    }
  }
}

__global__ void k_fill_u8(unsigned char* p, long long n, unsigned char v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void k_copy_i64(const int64_t* __restrict__ in, long long n, int64_t* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}

extern "C" size_t pcc_topk_ws_bytes(int64_t n) {
  if (n <= 0) return SEL_WS_HDR;
  return SEL_WS_HDR + 2 * pcc_align_up((size_t)pcc_cdiv(n, SEL_B) * sizeof(int2)) + 256;
}

// keys / keys_out nullable: the mask alone (pcc_topk_mask).  keys_out receives the kept rows' keys batch after batch
// (batch b starts at the sum of min(k, rows) of the batches before it -- sizes the host has).
static int topk_impl(const float* logits, int64_t stride_elems, const int64_t* h_seg_begin, const int64_t* h_k, int32_t nb,
                     const int64_t* keys, uint8_t* mask, int64_t* keys_out, void* ws, size_t ws_bytes, hipStream_t s) {
  PCC_REQUIRE(h_seg_begin && h_k && nb >= 0, "pcc_topk: bad arguments");
  int64_t out_base = 0;
  for (int b = 0; b < nb; ++b) {
    const int64_t b0 = h_seg_begin[b], n = h_seg_begin[b + 1] - b0, k = h_k[b];
    if (n <= 0) continue;
    PCC_REQUIRE(logits && mask && stride_elems >= 1, "pcc_topk: NULL array");
    PCC_REQUIRE(n < (1ll << 31), "pcc_topk: too many rows");
    const unsigned g = (unsigned)pcc_cdiv(n, 256);
    if (k <= 0 || k >= n) {
      k_fill_u8<<<g, 256, 0, s>>>(mask + b0, n, k >= n ? 1 : 0);
      PCC_LAUNCH_CHECK();
      if (k >= n && keys_out) {
        k_copy_i64<<<g, 256, 0, s>>>(keys + b0, n, keys_out + out_base);
        PCC_LAUNCH_CHECK();
      }
      if (k >= n) out_base += n;
      continue;
    }
    if (ws_bytes < pcc_topk_ws_bytes(n)) {
      pcc_set_error("pcc_topk: workspace too small");
      return PCC_EWS;
    }
    SelState* st = (SelState*)ws;
    int* hist = (int*)((char*)ws + 256);
    const int nblk = (int)pcc_cdiv(n, SEL_B);
    int2* cnt = (int2*)((char*)ws + SEL_WS_HDR);
    int2* before = (int2*)((char*)cnt + pcc_align_up((size_t)nblk * sizeof(int2)));
    const float* lg = logits + b0 * stride_elems;
    k_sel_init<<<1, 256, 0, s>>>(st, hist, (int)k);
    PCC_LAUNCH_CHECK();
    const unsigned gh = g < 1024 ? g : 1024;
    k_sel_hist<11><<<gh, 256, 0, s>>>(lg, stride_elems, n, 21, st, hist);
    k_sel_pick<<<1, 256, 0, s>>>(st, hist, 21, 11);
    k_sel_hist<11><<<gh, 256, 0, s>>>(lg, stride_elems, n, 10, st, hist);
    k_sel_pick<<<1, 256, 0, s>>>(st, hist, 10, 11);
    k_sel_hist<10><<<gh, 256, 0, s>>>(lg, stride_elems, n, 0, st, hist);
    k_sel_pick<<<1, 256, 0, s>>>(st, hist, 0, 10);
    PCC_LAUNCH_CHECK();
    k_sel_count<<<(unsigned)nblk, 256, 0, s>>>(lg, stride_elems, n, st, cnt);
    k_sel_offsets<<<1, 256, 0, s>>>(cnt, nblk, st, before);
    k_sel_compact<<<(unsigned)nblk, 256, 0, s>>>(lg, stride_elems, n, st, before, keys ? keys + b0 : nullptr, mask + b0,
                                                keys_out ? keys_out + out_base : nullptr);
    PCC_LAUNCH_CHECK();
    out_base += k;
  }
  return PCC_OK;
}

extern "C" int pcc_topk_mask(const float* logits, int64_t stride_elems, const int64_t* h_seg_begin,
                             const int64_t* h_k, int32_t nb, uint8_t* mask, void* ws, size_t ws_bytes,
                             void* stream) {
  return topk_impl(logits, stride_elems, h_seg_begin, h_k, nb, nullptr, mask, nullptr, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int pcc_topk_prune_keys(const float* logits, int64_t stride_elems, const int64_t* h_seg_begin,
                                   const int64_t* h_k, int32_t nb, const int64_t* keys, uint8_t* mask, int64_t* keys_out,
                                   void* ws, size_t ws_bytes, void* stream) {
  PCC_REQUIRE(keys && keys_out, "pcc_topk_prune_keys: NULL array");
  return topk_impl(logits, stride_elems, h_seg_begin, h_k, nb, keys, mask, keys_out, ws, ws_bytes, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// pruning: stable compaction of (key, feature row) by mask
// ------------------------------------------------------------------------------------------
__global__ void k_mask_to_int(const unsigned char* __restrict__ m, long long n, int* __restrict__ f) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f[i] = m[i] ? 1 : 0;
}

__global__ void k_prune_keys(const unsigned char* __restrict__ m, long long n, const int* __restrict__ pos,
                             const int64_t* __restrict__ keys, int64_t* __restrict__ keys_out,
                             int64_t* __restrict__ d_count) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (m[i] && keys) keys_out[pos[i]] = keys[i];
  if (i == n - 1 && d_count) *d_count = (int64_t)pos[i] + (m[i] ? 1 : 0);
}

template <typename VT>
__global__ void k_prune_feat(const unsigned char* __restrict__ m, long long n, const int* __restrict__ pos,
                             const VT* __restrict__ feat, int vpr, VT* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long r = t / vpr;
  if (r >= n) return;
  if (m[r]) out[(long long)pos[r] * vpr + (t - r * vpr)] = feat[t];
}

extern "C" size_t pcc_prune_ws_bytes(int64_t n) {
  if (n <= 0) return 256;
  return pcc_align_up((size_t)n * 4) + pcc_scan_ws_bytes(n) + 256;
}

extern "C" int pcc_prune_rows(const uint8_t* mask, int64_t n, const int64_t* keys, const float* feat, int32_t c,
                              int64_t* keys_out, float* feat_out, int64_t* d_count, void* ws, size_t ws_bytes,
                              void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n <= 0) {
    if (d_count) PCC_CHECK_HIP(hipMemsetAsync(d_count, 0, sizeof(int64_t), s));
    return PCC_OK;
  }
  PCC_REQUIRE(mask && ws, "pcc_prune_rows: NULL array");
  PCC_REQUIRE(n < (1ll << 31), "pcc_prune_rows: too many rows");
  if (ws_bytes < pcc_prune_ws_bytes(n)) {
    pcc_set_error("pcc_prune_rows: workspace too small");
    return PCC_EWS;
  }
  int* pos = (int*)ws;
  char* p = (char*)ws + pcc_align_up((size_t)n * 4);
  const unsigned g = (unsigned)pcc_cdiv(n, 256);
  k_mask_to_int<<<g, 256, 0, s>>>(mask, n, pos);
  PCC_LAUNCH_CHECK();
  PCC_TRY(pcc_scan_exclusive_i32(pos, pos, n, p, ws_bytes - (size_t)(p - (char*)ws), s));
  k_prune_keys<<<g, 256, 0, s>>>(mask, n, pos, keys, keys_out, d_count);
  PCC_LAUNCH_CHECK();
  if (feat && c > 0) {
    PCC_REQUIRE(feat_out, "pcc_prune_rows: feat_out is NULL");
    if (c % 4 == 0) {
      const int vpr = c / 4;
      k_prune_feat<float4><<<(unsigned)pcc_cdiv(n * vpr, 256), 256, 0, s>>>(mask, n, pos, (const float4*)feat, vpr,
                                                                          (float4*)feat_out);
    } else {
      k_prune_feat<float><<<(unsigned)pcc_cdiv(n * c, 256), 256, 0, s>>>(mask, n, pos, feat, c, feat_out);
    }
    PCC_LAUNCH_CHECK();
  }
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// exact-match lookup / gather
// ------------------------------------------------------------------------------------------
__global__ void k_lookup_rows(const int64_t* __restrict__ keys, int n, const int64_t* __restrict__ q, long long nq,
                              int* __restrict__ rows) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) rows[i] = pcc_find(keys, n, q[i]);
}

extern "C" int pcc_lookup_rows(const int64_t* keys, int64_t n, const int64_t* query_keys, int64_t nq,
                               int32_t* rows_out, void* stream) {
  if (nq <= 0) return PCC_OK;
  PCC_REQUIRE(query_keys && rows_out && (n == 0 || keys) && n < (1ll << 31), "pcc_lookup_rows: bad arguments");
  k_lookup_rows<<<(unsigned)pcc_cdiv(nq, 256), 256, 0, (hipStream_t)stream>>>(keys, (int)n, query_keys, nq, rows_out);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// one wave resolves 64 queries (one search per lane), then copies the 64 rows cooperatively
__global__ void __launch_bounds__(256) k_lookup_gather(const int64_t* __restrict__ keys, int n,
                                                       const float* __restrict__ feat, int c,
                                                       const int64_t* __restrict__ q, long long nq,
                                                       float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long q0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (q0 >= nq) return;
  const long long qi = q0 + lane;
  const int idx = (qi < nq) ? pcc_find(keys, n, q[qi]) : -1;
  const int cnt = (int)min(64ll, nq - q0);
  for (int j = 0; j < cnt; ++j) {
    const int src = __shfl(idx, j);
    float* dst = out + (q0 + j) * c;
    if (src >= 0) {
      const float* sp = feat + (long long)src * c;
      for (int e = lane; e < c; e += 64) dst[e] = sp[e];
    } else {
      for (int e = lane; e < c; e += 64) dst[e] = 0.f;
    }
  }
}

extern "C" int pcc_lookup_gather(const int64_t* keys, int64_t n, const float* feat, int32_t c,
                                 const int64_t* query_keys, int64_t nq, float* out, void* stream) {
  if (nq <= 0) return PCC_OK;
  PCC_REQUIRE(query_keys && out && c >= 1 && (n == 0 || (keys && feat)) && n < (1ll << 31),
              "pcc_lookup_gather: bad arguments");
  k_lookup_gather<<<(unsigned)pcc_cdiv(nq, 256), 256, 0, (hipStream_t)stream>>>(keys, (int)n, feat, c, query_keys, nq,
                                                                                 out);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}
