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

__global__ void k_sel_init(SelState* st, int* hist, int k) {
  if (threadIdx.x == 0) { st->prefix = 0; st->mask = 0; st->remaining = k; st->pad = 0; }
  hist[threadIdx.x] = 0;
}

__global__ void __launch_bounds__(256) k_sel_hist(const float* __restrict__ logits, long long stride, long long n,
                                                  int shift, const SelState* __restrict__ st, int* __restrict__ hist) {
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const unsigned prefix = st->prefix, mask = st->mask;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const unsigned u = f2u(logits[i * stride]);
    if ((u & mask) == prefix) atomicAdd(&h[(u >> shift) & 0xFF], 1);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// pick the digit holding the `remaining`-th largest candidate; clears hist for the next pass
__global__ void k_sel_pick(SelState* st, int* hist, int shift) {
  __shared__ int h[256];
  h[threadIdx.x] = hist[threadIdx.x];
  hist[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    int rem = st->remaining, cum = 0, d = 255;
    for (; d > 0; --d) {
      if (cum + h[d] >= rem) break;
      cum += h[d];
    }
    st->remaining = rem - cum;
    st->prefix |= (unsigned)d << shift;
    st->mask |= 0xFFu << shift;
  }
}

__global__ void k_sel_eqflags(const float* __restrict__ logits, long long stride, long long n,
                              const SelState* __restrict__ st, int* __restrict__ eq) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  eq[i] = (f2u(logits[i * stride]) == st->prefix) ? 1 : 0;
}

__global__ void k_sel_mask(const float* __restrict__ logits, long long stride, long long n,
                           const SelState* __restrict__ st, const int* __restrict__ eq_rank,
                           unsigned char* __restrict__ mask) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned u = f2u(logits[i * stride]), T = st->prefix;
  mask[i] = (u > T || (u == T && eq_rank[i] < st->remaining)) ? 1 : 0;
}

__global__ void k_fill_u8(unsigned char* p, long long n, unsigned char v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

extern "C" size_t pcc_topk_ws_bytes(int64_t n) {
  if (n <= 0) return 2048;
  return 2048 + pcc_align_up((size_t)n * 4) + pcc_scan_ws_bytes(n);
}

extern "C" int pcc_topk_mask(const float* logits, int64_t stride_elems, const int64_t* h_seg_begin,
                             const int64_t* h_k, int32_t nb, uint8_t* mask, void* ws, size_t ws_bytes,
                             void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(h_seg_begin && h_k && nb >= 0, "pcc_topk_mask: bad arguments");
  for (int b = 0; b < nb; ++b) {
    const int64_t b0 = h_seg_begin[b], n = h_seg_begin[b + 1] - b0, k = h_k[b];
    if (n <= 0) continue;
    PCC_REQUIRE(logits && mask && stride_elems >= 1, "pcc_topk_mask: NULL array");
    PCC_REQUIRE(n < (1ll << 31), "pcc_topk_mask: too many rows");
    const unsigned g = (unsigned)pcc_cdiv(n, 256);
    if (k <= 0 || k >= n) {
      k_fill_u8<<<g, 256, 0, s>>>(mask + b0, n, k >= n ? 1 : 0);
      PCC_LAUNCH_CHECK();
      continue;
    }
    if (ws_bytes < pcc_topk_ws_bytes(n)) {
      pcc_set_error("pcc_topk_mask: workspace too small");
      return PCC_EWS;
    }
    SelState* st = (SelState*)ws;
    int* hist = (int*)((char*)ws + 256);
    int* eq = (int*)((char*)ws + 2048);
    void* scan_ws = (char*)eq + pcc_align_up((size_t)n * 4);
    const float* lg = logits + b0 * stride_elems;
    k_sel_init<<<1, 256, 0, s>>>(st, hist, (int)k);
    PCC_LAUNCH_CHECK();
    const unsigned gh = g < 1024 ? g : 1024;
    for (int shift = 24; shift >= 0; shift -= 8) {
      k_sel_hist<<<gh, 256, 0, s>>>(lg, stride_elems, n, shift, st, hist);
      PCC_LAUNCH_CHECK();
      k_sel_pick<<<1, 256, 0, s>>>(st, hist, shift);
      PCC_LAUNCH_CHECK();
    }
    k_sel_eqflags<<<g, 256, 0, s>>>(lg, stride_elems, n, st, eq);
    PCC_LAUNCH_CHECK();
    PCC_TRY(pcc_scan_exclusive_i32(eq, eq, n, scan_ws, ws_bytes - 2048 - pcc_align_up((size_t)n * 4), s));
    k_sel_mask<<<g, 256, 0, s>>>(lg, stride_elems, n, st, eq, mask + b0);
    PCC_LAUNCH_CHECK();
  }
  return PCC_OK;
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
