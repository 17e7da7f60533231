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
// Round 4: ALL batch segments of a call go through the same launches (blockIdx.y = segment; a training batch of 4 cubes x 3
// levels was 130 launches per step, ~11 per segment).  The arithmetic per segment is unchanged.
static constexpr int SEL_MAXBINS = 2048;
static constexpr int SEL_B = 2048;                                    // rows per workgroup of the count / compact passes (256 x 8)
static constexpr int SEL_MAXSEG = 48;                                 // segments per batch of launches (table passed by value)
static constexpr int SEL_SEG_WS = 256 + SEL_MAXBINS * 4;              // per segment: state | histogram

// mode: 0 = select the k largest, 1 = keep every row (k >= n), 2 = keep none (k <= 0)
struct SelSeg { long long b0, out_base; int n, k, mode, blk0; };
struct SelTable { int nseg, pad; SelSeg s[SEL_MAXSEG]; };

__device__ inline SelState* sel_state(void* ws, int seg) { return (SelState*)((char*)ws + (size_t)seg * SEL_SEG_WS); }
__device__ inline int* sel_hist(void* ws, int seg) { return (int*)((char*)ws + (size_t)seg * SEL_SEG_WS + 256); }

__global__ void k_sel_init(void* ws, SelTable t) {
  const int seg = blockIdx.y;
  SelState* st = sel_state(ws, seg);
  int* hist = sel_hist(ws, seg);
  if (threadIdx.x == 0) { st->prefix = 0; st->mask = 0; st->remaining = t.s[seg].k; st->pad = 0; }
  for (int i = threadIdx.x; i < SEL_MAXBINS; i += blockDim.x) hist[i] = 0;
}

template <int BITS>
__global__ void __launch_bounds__(256) k_sel_hist(const float* __restrict__ logits, long long stride, int shift, void* ws, SelTable t) {
  constexpr int NB = 1 << BITS;
  __shared__ int h[NB];
  const int seg = blockIdx.y;
  const SelSeg sg = t.s[seg];
  if (sg.mode != 0 || (long long)blockIdx.x * 256 >= sg.n) return;
  for (int i = threadIdx.x; i < NB; i += 256) h[i] = 0;
  __syncthreads();
  const SelState* st = sel_state(ws, seg);
  int* hist = sel_hist(ws, seg);
  const unsigned prefix = st->prefix, mask = st->mask;
  const float* lg = logits + sg.b0 * stride;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < sg.n; i += (long long)gridDim.x * 256) {
    const unsigned u = f2u(lg[i * stride]);
    if ((u & mask) == prefix) atomicAdd(&h[(u >> shift) & (NB - 1)], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NB; i += 256)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

// pick the digit holding the `remaining`-th largest candidate (bins walked from the top); clears hist for the next pass.
// 256 threads: thread t owns the NB/256 bins below NB - t * per; an exclusive scan over the threads finds the owner.
__global__ void __launch_bounds__(256) k_sel_pick(void* ws, int shift, int bits, SelTable t) {
  __shared__ int wsum[4];
  __shared__ int h[SEL_MAXBINS];
  const int seg = blockIdx.y;
  if (t.s[seg].mode != 0) return;
  SelState* st = sel_state(ws, seg);
  int* hist = sel_hist(ws, seg);
  const int nb = 1 << bits, per = nb >> 8;
  for (int i = threadIdx.x; i < nb; i += 256) { h[i] = hist[i]; hist[i] = 0; }
  __syncthreads();
  const int top = nb - 1 - (int)threadIdx.x * per;                    // this thread's bins: top, top-1, ..., top-per+1
  int mine = 0;
  for (int q = 0; q < per; ++q) mine += h[top - q];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int tt = __shfl_up(inc, d); if (lane >= d) inc += tt; }
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

__global__ void __launch_bounds__(256) k_sel_count(const float* __restrict__ logits, long long stride, void* ws, SelTable t,
                                                   int2* __restrict__ cnt) {
  __shared__ int wsum[4];
  const int seg = blockIdx.y;
  const SelSeg sg = t.s[seg];
  if (sg.mode != 0 || (long long)blockIdx.x * SEL_B >= sg.n) return;
  const unsigned T = sel_state(ws, seg)->prefix;
  const float* lg = logits + sg.b0 * stride;
  const long long base = (long long)blockIdx.x * SEL_B + (long long)threadIdx.x * 8;
  int gt = 0, eq = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (base + q < sg.n) {
      const unsigned u = f2u(lg[(base + q) * stride]);
      gt += u > T; eq += u == T;
    }
  int tg, te;
  sel_block_scan(gt, &tg, wsum);
  sel_block_scan(eq, &te, wsum);
  if (threadIdx.x == 0) cnt[sg.blk0 + blockIdx.x] = make_int2(tg, te);
}

// one workgroup per segment: per counted workgroup b the equal values before it and the rows kept before it (exclusive scans)
__global__ void __launch_bounds__(256) k_sel_offsets(const int2* __restrict__ cnt, void* ws, SelTable t, int2* __restrict__ before) {
  __shared__ int wsum[4];
  const int seg = blockIdx.y;
  const SelSeg sg = t.s[seg];
  if (sg.mode != 0) return;
  const int nblk = (sg.n + SEL_B - 1) / SEL_B;
  const int rem = sel_state(ws, seg)->remaining;
  int carry_eq = 0, carry_keep = 0;
  for (int b0 = 0; b0 < nblk; b0 += 256) {
    const int b = b0 + (int)threadIdx.x;
    const int2 c = b < nblk ? cnt[sg.blk0 + b] : make_int2(0, 0);
    int tot;
    const int eqb = carry_eq + sel_block_scan(c.y, &tot, wsum);
    carry_eq += tot;
    const int room = rem - eqb;
    const int kept = c.x + (room <= 0 ? 0 : (room < c.y ? room : c.y));
    const int keepb = carry_keep + sel_block_scan(kept, &tot, wsum);
    carry_keep += tot;
    if (b < nblk) before[sg.blk0 + b] = make_int2(eqb, keepb);
  }
}

__global__ void __launch_bounds__(256) k_sel_compact(const float* __restrict__ logits, long long stride, void* ws, SelTable t,
                                                     const int2* __restrict__ before, const int64_t* __restrict__ keys,
                                                     unsigned char* __restrict__ mask, int64_t* __restrict__ keys_out) {
  __shared__ int wsum[4];
  const int seg = blockIdx.y;
  const SelSeg sg = t.s[seg];
  if ((long long)blockIdx.x * SEL_B >= sg.n) return;
  const long long base = (long long)blockIdx.x * SEL_B + (long long)threadIdx.x * 8;
  if (sg.mode != 0) {                                                  // every row / no row
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (base + q < sg.n) {
        mask[sg.b0 + base + q] = sg.mode == 1 ? 1 : 0;
        if (sg.mode == 1 && keys_out) keys_out[sg.out_base + base + q] = keys[sg.b0 + base + q];
      }
    return;
  }
  const SelState* st = sel_state(ws, seg);
  const unsigned T = st->prefix;
  const int rem = st->remaining;
  const int2 bf = before[sg.blk0 + blockIdx.x];
  const float* lg = logits + sg.b0 * stride;
  unsigned u[8];
  int eq = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    u[q] = base + q < sg.n ? f2u(lg[(base + q) * stride]) : 0u;
    eq += (base + q < sg.n) && u[q] == T;
  }
  int tot;
  int eq_rank = bf.x + sel_block_scan(eq, &tot, wsum);
  bool keep[8];
  int kc = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const bool in = base + q < sg.n;
    const bool e = in && u[q] == T;
    keep[q] = in && (u[q] > T || (e && eq_rank < rem));
    eq_rank += e;
    kc += keep[q];
  }
  long long pos = sg.out_base + bf.y + sel_block_scan(kc, &tot, wsum);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (base + q >= sg.n) break;
    mask[sg.b0 + base + q] = keep[q] ? 1 : 0;
    if (keep[q]) {
      if (keys_out) keys_out[pos] = keys[sg.b0 + base + q];
      ++pos;
    }
  }
}

extern "C" size_t pcc_topk_ws_bytes(int64_t n) {
  if (n < 0) n = 0;
  return (size_t)SEL_MAXSEG * SEL_SEG_WS + 2 * pcc_align_up((size_t)(pcc_cdiv(n, SEL_B) + SEL_MAXSEG) * sizeof(int2)) + 256;
}

// keys / keys_out nullable: the mask alone (pcc_topk_mask).  keys_out receives the kept rows' keys batch after batch
// (batch b starts at the sum of min(k, rows) of the batches before it -- sizes the host has).
static int topk_impl(const float* logits, int64_t stride_elems, const int64_t* h_seg_begin, const int64_t* h_k, int32_t nb,
                     const int64_t* keys, uint8_t* mask, int64_t* keys_out, void* ws, size_t ws_bytes, hipStream_t s) {
  PCC_REQUIRE(h_seg_begin && h_k && nb >= 0, "pcc_topk: bad arguments");
  int64_t out_base = 0;
  int b = 0;
  while (b < nb) {
    SelTable t;
    t.nseg = 0; t.pad = 0;
    int blk = 0;
    long long max_n = 0;
    bool any_select = false;
    for (; b < nb && t.nseg < SEL_MAXSEG; ++b) {
      const int64_t b0 = h_seg_begin[b], n = h_seg_begin[b + 1] - b0, k = h_k[b];
      if (n <= 0) continue;
      PCC_REQUIRE(logits && mask && stride_elems >= 1, "pcc_topk: NULL array");
      PCC_REQUIRE(n < (1ll << 31), "pcc_topk: too many rows");
      SelSeg& sg = t.s[t.nseg++];
      sg.b0 = b0; sg.out_base = out_base; sg.n = (int)n;
      sg.mode = k >= n ? 1 : (k <= 0 ? 2 : 0);
      sg.k = (int)(k < 0 ? 0 : (k > n ? n : k));
      sg.blk0 = blk;
      blk += (int)pcc_cdiv(n, SEL_B);
      if (sg.mode == 0) any_select = true;
      if (n > max_n) max_n = n;
      out_base += sg.k;
    }
    if (t.nseg == 0) continue;
    const size_t need = (size_t)SEL_MAXSEG * SEL_SEG_WS + 2 * pcc_align_up((size_t)blk * sizeof(int2));
    if (ws_bytes < need) {
      pcc_set_error("pcc_topk: workspace too small");
      return PCC_EWS;
    }
    int2* cnt = (int2*)((char*)ws + (size_t)SEL_MAXSEG * SEL_SEG_WS);
    int2* before = (int2*)((char*)cnt + pcc_align_up((size_t)blk * sizeof(int2)));
    const unsigned ny = (unsigned)t.nseg;
    const unsigned g = (unsigned)pcc_cdiv(max_n, 256);
    const unsigned gh = g < 1024 ? g : 1024;
    const unsigned gb = (unsigned)pcc_cdiv(max_n, SEL_B);
    if (any_select) {
      k_sel_init<<<dim3(1, ny), 256, 0, s>>>(ws, t);
      k_sel_hist<11><<<dim3(gh, ny), 256, 0, s>>>(logits, stride_elems, 21, ws, t);
      k_sel_pick<<<dim3(1, ny), 256, 0, s>>>(ws, 21, 11, t);
      k_sel_hist<11><<<dim3(gh, ny), 256, 0, s>>>(logits, stride_elems, 10, ws, t);
      k_sel_pick<<<dim3(1, ny), 256, 0, s>>>(ws, 10, 11, t);
      k_sel_hist<10><<<dim3(gh, ny), 256, 0, s>>>(logits, stride_elems, 0, ws, t);
      k_sel_pick<<<dim3(1, ny), 256, 0, s>>>(ws, 0, 10, t);
      PCC_LAUNCH_CHECK();
      k_sel_count<<<dim3(gb, ny), 256, 0, s>>>(logits, stride_elems, ws, t, cnt);
      k_sel_offsets<<<dim3(1, ny), 256, 0, s>>>(cnt, ws, t, before);
    }
    k_sel_compact<<<dim3(gb, ny), 256, 0, s>>>(logits, stride_elems, ws, t, before, keys, mask, keys_out);
    PCC_LAUNCH_CHECK();
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
