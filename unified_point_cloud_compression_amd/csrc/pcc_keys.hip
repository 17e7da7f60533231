// Coordinate keys: pack/unpack, exclusive scan, LSD radix sort, adjacent unique, stride and
// generative expansion of canonical coordinate sets.  All HBM-bound integer work:
// 16-byte coalesced accesses, wave ballots for ranking, no atomics on global memory.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "pcc_common.h"

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void pcc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* pcc_last_error(void) { return g_err; }
extern "C" int pcc_version(void) { return 100; }

extern "C" int pcc_device_info(int* h_cu_count, char* h_arch, int h_arch_len) {
  int dev = 0;
  PCC_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  PCC_CHECK_HIP(hipGetDeviceProperties(&p, dev));
  if (h_cu_count) *h_cu_count = p.multiProcessorCount;
  if (h_arch && h_arch_len > 0) {
    strncpy(h_arch, p.gcnArchName, h_arch_len - 1);
    h_arch[h_arch_len - 1] = 0;
  }
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// pack / unpack
// ------------------------------------------------------------------------------------------
__device__ inline int64_t pack4(int b, int x, int y, int z) {
  return ((int64_t)b << 48) | ((int64_t)(x + PCC_BIAS) << 32) | ((int64_t)(y + PCC_BIAS) << 16) |
         (int64_t)(z + PCC_BIAS);
}

__global__ void k_pack_i32(const int4* __restrict__ c, int64_t n, int64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 v = c[i];
  keys[i] = pack4(v.x, v.y, v.z, v.w);
}

__global__ void k_pack_f32(const float4* __restrict__ c, int64_t n, int64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = c[i];
  keys[i] = pack4((int)floorf(v.x), (int)floorf(v.y), (int)floorf(v.z), (int)floorf(v.w));
}

__global__ void k_unpack(const int64_t* __restrict__ keys, int64_t n, int4* __restrict__ c) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t k = keys[i];
  int4 v;
  v.x = (int)(k >> 48);
  v.y = (int)((k >> 32) & 0xFFFF) - (int)PCC_BIAS;
  v.z = (int)((k >> 16) & 0xFFFF) - (int)PCC_BIAS;
  v.w = (int)(k & 0xFFFF) - (int)PCC_BIAS;
  c[i] = v;
}

static inline dim3 grid1(int64_t n, int bs = 256) { return dim3((unsigned)pcc_cdiv(n, bs)); }

extern "C" int pcc_keys_pack_i32(const int32_t* coords, int64_t n, int64_t* keys, void* stream) {
  if (n == 0) return PCC_OK;
  PCC_REQUIRE(coords && keys && n > 0, "pcc_keys_pack_i32: bad arguments");
  k_pack_i32<<<grid1(n), 256, 0, (hipStream_t)stream>>>((const int4*)coords, n, keys);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_keys_pack_f32(const float* coords, int64_t n, int64_t* keys, void* stream) {
  if (n == 0) return PCC_OK;
  PCC_REQUIRE(coords && keys && n > 0, "pcc_keys_pack_f32: bad arguments");
  k_pack_f32<<<grid1(n), 256, 0, (hipStream_t)stream>>>((const float4*)coords, n, keys);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_keys_unpack(const int64_t* keys, int64_t n, int32_t* coords, void* stream) {
  if (n == 0) return PCC_OK;
  PCC_REQUIRE(coords && keys && n > 0, "pcc_keys_unpack: bad arguments");
  k_unpack<<<grid1(n), 256, 0, (hipStream_t)stream>>>(keys, n, (int4*)coords);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// Row ranges per batch index of a canonical key array whose row count is still on the device: out[e] = first row whose batch
// index is >= e, e in [0, entries) (entries = batches + 1 <= 12: the last one is the row count).  Queued behind the kernel that
// produces the set, so the ranges come back with the set's size in the SAME host read (`model/transforms.py:228-254` selects
// the top-k per batch; the training step read the ranges of every level's candidate set on their own).
__global__ void k_batch_bounds(const long long* __restrict__ keys, const long long* __restrict__ d_n, long long n_host, int entries,
                               long long* __restrict__ out_a, long long* __restrict__ out_b, long long* __restrict__ out_c) {
  const int e = threadIdx.x;
  if (e >= entries) return;
  const long long n = d_n ? *d_n : n_host, target = (long long)e << 48;
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  if (e < 4) out_a[e] = lo; else if (e < 8) out_b[e - 4] = lo; else out_c[e - 8] = lo;
}

extern "C" int pcc_batch_bounds(const int64_t* keys, const int64_t* d_n, int64_t n_host, int32_t entries, int64_t* out_a,
                                int64_t* out_b, int64_t* out_c, void* stream) {
  PCC_REQUIRE(keys && out_a && n_host >= 0 && entries >= 1 && entries <= 12 && (entries <= 4 || out_b) && (entries <= 8 || out_c),
              "pcc_batch_bounds: bad arguments");
  k_batch_bounds<<<1, 64, 0, (hipStream_t)stream>>>((const long long*)keys, (const long long*)d_n, n_host, entries,
                                                    (long long*)out_a, (long long*)out_b, (long long*)out_c);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// exclusive scan (int32), 256 threads x 8 items per block, recursive over block sums
// ------------------------------------------------------------------------------------------
static constexpr int SCAN_T = 256;
static constexpr int SCAN_I = 8;
static constexpr int SCAN_B = SCAN_T * SCAN_I;  // 2048

// block-wide exclusive scan of one value per thread; returns exclusive prefix, total in *total
__device__ inline int block_excl_scan(int v, int* total) {
  __shared__ int wsum[SCAN_T / PCC_WAVE];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_T / PCC_WAVE; ++i) {
    const int s = wsum[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// A thread's 8 consecutive items: two 16-byte accesses when the array is 16-byte aligned and the items are inside it (the
// element-wise form costs 8 accesses of 4 bytes at a 32-byte lane pitch -- 16 cache lines per wave instruction; the large scans
// of a step, 14.5 M candidates' list starts and masks, ran at a third of the memory rate).
__device__ __forceinline__ void scan_load8(const int* __restrict__ in, int64_t n, int64_t base, int (&v)[SCAN_I]) {
  static_assert(SCAN_I == 8, "two int4 per thread");
  if (base + SCAN_I <= n && ((uintptr_t)in & 15) == 0) {
    const int4 a = reinterpret_cast<const int4*>(in + base)[0], b = reinterpret_cast<const int4*>(in + base)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) v[i] = (base + i < n) ? in[base + i] : 0;
  }
}
__device__ __forceinline__ void scan_store8(int* __restrict__ out, int64_t n, int64_t base, int run, const int (&v)[SCAN_I]) {
  int o[SCAN_I];
#pragma unroll
  for (int i = 0; i < SCAN_I; ++i) { o[i] = run; run += v[i]; }
  if (base + SCAN_I <= n && ((uintptr_t)out & 15) == 0) {
    reinterpret_cast<int4*>(out + base)[0] = make_int4(o[0], o[1], o[2], o[3]);
    reinterpret_cast<int4*>(out + base)[1] = make_int4(o[4], o[5], o[6], o[7]);
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i)
      if (base + i < n) out[base + i] = o[i];
  }
}

__global__ void __launch_bounds__(SCAN_T) k_scan_reduce(const int* __restrict__ in, int64_t n,
                                                        int* __restrict__ sums) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_I;
  int v[SCAN_I];
  scan_load8(in, n, base, v);
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_I; ++i) s += v[i];
  int tot;
  block_excl_scan(s, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_T) k_scan_apply(const int* __restrict__ in, int64_t n,
                                                       const int* __restrict__ offs,
                                                       int* __restrict__ out) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_I;
  int v[SCAN_I];
  scan_load8(in, n, base, v);
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_I; ++i) s += v[i];
  int tot;
  const int run = block_excl_scan(s, &tot) + (offs ? offs[blockIdx.x] : 0);
  scan_store8(out, n, base, run, v);
}

// apply pass that derives its block's offset from the raw block totals itself (offset = sum of the totals of the blocks
// before it: <= SCAN_DIRECT_NB ints, all L2 hits): reduce + apply, two launches instead of the three to five of the
// recursive form (a step runs ~35 scans; at ~8 us of launch latency each level mattered more than its work)
static constexpr int64_t SCAN_DIRECT_NB = 16384;
__global__ void __launch_bounds__(SCAN_T) k_scan_apply_direct(const int* __restrict__ in, int64_t n,
                                                              const int* __restrict__ totals, int* __restrict__ out) {
  __shared__ int s_off;
  int part = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += SCAN_T) part += totals[i];
  int tot;
  block_excl_scan(part, &tot);
  if (threadIdx.x == 0) s_off = tot;
  __syncthreads();
  const int off = s_off;
  const int64_t base = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_I;
  int v[SCAN_I];
  scan_load8(in, n, base, v);
  int sum = 0;
#pragma unroll
  for (int i = 0; i < SCAN_I; ++i) sum += v[i];
  const int run = block_excl_scan(sum, &tot) + off;
  scan_store8(out, n, base, run, v);
}

size_t pcc_scan_ws_bytes(int64_t n) {
  size_t tot = 0;
  int64_t m = n;
  while (m > SCAN_B) {
    m = pcc_cdiv(m, SCAN_B);
    tot += pcc_align_up((size_t)m * sizeof(int));
  }
  return tot + 256;
}

int pcc_scan_exclusive_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes,
                           hipStream_t s) {
  if (n <= 0) return PCC_OK;
  if (ws_bytes < pcc_scan_ws_bytes(n)) {
    pcc_set_error("scan: workspace too small");
    return PCC_EWS;
  }
  const int64_t nb = pcc_cdiv(n, SCAN_B);
  if (nb == 1) {
    k_scan_apply<<<1, SCAN_T, 0, s>>>(in, n, nullptr, out);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
  }
  int* sums = (int*)ws;
  const size_t used = pcc_align_up((size_t)nb * sizeof(int));
  k_scan_reduce<<<(unsigned)nb, SCAN_T, 0, s>>>(in, n, sums);
  PCC_LAUNCH_CHECK();
  if (nb <= SCAN_DIRECT_NB) {
    k_scan_apply_direct<<<(unsigned)nb, SCAN_T, 0, s>>>(in, n, sums, out);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
  }
  PCC_TRY(pcc_scan_exclusive_i32(sums, sums, nb, (char*)ws + used, ws_bytes - used, s));
  k_scan_apply<<<(unsigned)nb, SCAN_T, 0, s>>>(in, n, sums, out);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// LSD radix sort, 8-bit digits, 2048 keys per block, stable.
// ------------------------------------------------------------------------------------------
static constexpr int RS_T = 256;
static constexpr int RS_I = 8;
static constexpr int RS_B = RS_T * RS_I;

template <typename KT>
__global__ void __launch_bounds__(RS_T) k_rs_hist(const KT* __restrict__ keys, int64_t n, int shift,
                                                  int nblocks, int* __restrict__ hist) {
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_B;
#pragma unroll
  for (int r = 0; r < RS_I; ++r) {
    const int64_t e = base + r * RS_T + threadIdx.x;
    if (e < n) atomicAdd(&h[(int)((keys[e] >> shift) & 0xFF)], 1);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

template <typename KT, bool PAYLOAD>
__global__ void __launch_bounds__(RS_T) k_rs_scatter(const KT* __restrict__ keys,
                                                     const int* __restrict__ pay_in, int64_t n, int shift,
                                                     int nblocks, const int* __restrict__ offs,
                                                     KT* __restrict__ keys_out,
                                                     int* __restrict__ pay_out) {
  constexpr int NW = RS_T / PCC_WAVE;           // 4 waves
  constexpr int NWR = NW * RS_I;                // 32 wave-rounds, in element order
  __shared__ unsigned short wcount[NWR][256];   // 16 KB
  __shared__ int gbase[256];
  __shared__ int dstart[256];
  __shared__ int wtot[NW];
  __shared__ KT skey[RS_B];
  __shared__ int spay[PAYLOAD ? RS_B : 1];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < NWR * 256 / 2; i += RS_T) ((unsigned*)wcount)[i] = 0u;
  gbase[tid] = offs[(int64_t)tid * nblocks + blockIdx.x];
  __syncthreads();

  const int64_t base = (int64_t)blockIdx.x * RS_B;
  KT key[RS_I];
  int rank[RS_I];
#pragma unroll
  for (int r = 0; r < RS_I; ++r) {
    const int64_t e = base + r * RS_T + tid;
    const bool valid = e < n;
    key[r] = valid ? keys[e] : 0;
    const int d = (int)((key[r] >> shift) & 0xFF);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1;
      const unsigned long long bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    rank[r] = __popcll(peers & lt);
    if (valid && rank[r] == 0) wcount[r * NW + w][d] = (unsigned short)__popcll(peers);
  }
  __syncthreads();
  int run = 0;
  {  // per digit: exclusive prefix over the 32 wave-rounds
#pragma unroll 4
    for (int i = 0; i < NWR; ++i) {
      const int c = wcount[i][tid];
      wcount[i][tid] = (unsigned short)run;
      run += c;
    }
  }
  // first position of every digit inside the block (exclusive scan of the 256 digit totals)
  {
    int v = run;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const int t = __shfl_up(v, dd, 64);
      if (lane >= dd) v += t;
    }
    if (lane == 63) wtot[w] = v;
    __syncthreads();
    int add = 0;
    for (int i = 0; i < w; ++i) add += wtot[i];
    dstart[tid] = v - run + add;
  }
  __syncthreads();
  // stage the block's elements in LDS in digit order, then write them out: neighbouring threads hold neighbouring
  // elements of one digit run, so the global stores are contiguous runs instead of 2048 scattered words
#pragma unroll
  for (int r = 0; r < RS_I; ++r) {
    const int64_t e = base + r * RS_T + tid;
    if (e < n) {
      const int d = (int)((key[r] >> shift) & 0xFF);
      const int lp = dstart[d] + wcount[r * NW + w][d] + rank[r];
      skey[lp] = key[r];
      if (PAYLOAD) spay[lp] = pay_in ? pay_in[e] : (int)e;
    }
  }
  __syncthreads();
  const int nv = (int)min((int64_t)RS_B, n - base);
#pragma unroll
  for (int r = 0; r < RS_I; ++r) {
    const int j = r * RS_T + tid;
    if (j < nv) {
      const KT k = skey[j];
      const int d = (int)((k >> shift) & 0xFF);
      const int64_t pos = (int64_t)gbase[d] + (j - dstart[d]);
      keys_out[pos] = k;
      if (PAYLOAD) pay_out[pos] = spay[j];
    }
  }
}

__global__ void k_copy_keys_iota(const int64_t* __restrict__ in, int64_t n, int64_t* __restrict__ out,
                                 int* __restrict__ perm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (out != in) out[i] = in[i];
  if (perm) perm[i] = (int)i;
}

extern "C" size_t pcc_sort_ws_bytes(int64_t n) {
  if (n <= 0) return 256;
  const int64_t nb = pcc_cdiv(n, RS_B);
  return pcc_align_up((size_t)n * 8) + pcc_align_up((size_t)n * 4) + pcc_align_up((size_t)nb * 256 * 4) +
         pcc_scan_ws_bytes(nb * 256) + 256;
}

extern "C" int pcc_sort_keys(const int64_t* keys_in, int64_t n, uint64_t bit_mask, int64_t* keys_out,
                             int32_t* perm_out, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(keys_in && keys_out && keys_in != keys_out, "pcc_sort_keys: bad arguments");
  PCC_REQUIRE(n < (1ll << 31), "pcc_sort_keys: n too large");
  if (ws_bytes < pcc_sort_ws_bytes(n)) {
    pcc_set_error("pcc_sort_keys: workspace too small");
    return PCC_EWS;
  }
  const int64_t nb = pcc_cdiv(n, RS_B);
  char* p = (char*)ws;
  uint64_t* tmp_k = (uint64_t*)p;  p += pcc_align_up((size_t)n * 8);
  int* tmp_p = (int*)p;            p += pcc_align_up((size_t)n * 4);
  int* hist = (int*)p;             p += pcc_align_up((size_t)nb * 256 * 4);
  void* scan_ws = p;
  const size_t scan_bytes = ws_bytes - (size_t)(p - (char*)ws);

  int shifts[8], np = 0;
  for (int d = 0; d < 8; ++d)
    if ((bit_mask >> (8 * d)) & 0xFF) shifts[np++] = 8 * d;
  if (np == 0) {
    k_copy_keys_iota<<<grid1(n), 256, 0, s>>>(keys_in, n, keys_out, perm_out);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
  }
  const uint64_t* src_k = (const uint64_t*)keys_in;
  const int* src_p = nullptr;
  for (int i = 0; i < np; ++i) {
    // last pass must land in keys_out / perm_out
    const bool to_out = ((np - 1 - i) % 2) == 0;
    uint64_t* dst_k = to_out ? (uint64_t*)keys_out : tmp_k;
    int* dst_p = perm_out ? (to_out ? perm_out : tmp_p) : nullptr;
    k_rs_hist<uint64_t><<<(unsigned)nb, RS_T, 0, s>>>(src_k, n, shifts[i], (int)nb, hist);
    PCC_LAUNCH_CHECK();
    PCC_TRY(pcc_scan_exclusive_i32(hist, hist, nb * 256, scan_ws, scan_bytes, s));
    if (perm_out)
      k_rs_scatter<uint64_t, true><<<(unsigned)nb, RS_T, 0, s>>>(src_k, src_p, n, shifts[i], (int)nb, hist, dst_k, dst_p);
    else
      k_rs_scatter<uint64_t, false><<<(unsigned)nb, RS_T, 0, s>>>(src_k, nullptr, n, shifts[i], (int)nb, hist, dst_k, nullptr);
    PCC_LAUNCH_CHECK();
    src_k = dst_k;
    src_p = dst_p;
  }
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// adjacent unique on sorted keys
// ------------------------------------------------------------------------------------------
__global__ void k_uniq_flags(const int64_t* __restrict__ k, int64_t n, int* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[i] = (i == 0 || k[i] != k[i - 1]) ? 1 : 0;
}

__global__ void k_uniq_scatter(const int64_t* __restrict__ k, int64_t n, const int* __restrict__ pos,
                               int64_t* __restrict__ uniq, int* __restrict__ first,
                               int64_t* __restrict__ d_count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool f = (i == 0 || k[i] != k[i - 1]);
  if (f) {
    uniq[pos[i]] = k[i];
    if (first) first[pos[i]] = (int)i;
  }
  if (i == n - 1) *d_count = (int64_t)pos[i] + (f ? 1 : 0);
}

extern "C" size_t pcc_unique_ws_bytes(int64_t n) {
  if (n <= 0) return 256;
  return pcc_align_up((size_t)n * 4) + pcc_scan_ws_bytes(n) + 256;
}

extern "C" int pcc_unique_sorted(const int64_t* sorted_keys, int64_t n, int64_t* uniq, int32_t* first,
                                 int64_t* d_count, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(d_count, "pcc_unique_sorted: d_count is NULL");
  if (n <= 0) {
    PCC_CHECK_HIP(hipMemsetAsync(d_count, 0, sizeof(int64_t), s));
    return PCC_OK;
  }
  PCC_REQUIRE(sorted_keys && uniq && sorted_keys != uniq, "pcc_unique_sorted: bad arguments");
  if (ws_bytes < pcc_unique_ws_bytes(n)) {
    pcc_set_error("pcc_unique_sorted: workspace too small");
    return PCC_EWS;
  }
  int* pos = (int*)ws;
  char* p = (char*)ws + pcc_align_up((size_t)n * 4);
  k_uniq_flags<<<grid1(n), 256, 0, s>>>(sorted_keys, n, pos);
  PCC_LAUNCH_CHECK();
  PCC_TRY(pcc_scan_exclusive_i32(pos, pos, n, p, ws_bytes - (size_t)(p - (char*)ws), s));
  k_uniq_scatter<<<grid1(n), 256, 0, s>>>(sorted_keys, n, pos, uniq, first, d_count);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

__global__ void k_is_canonical(const int64_t* __restrict__ k, int64_t n, int* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 || i >= n) return;
  if (k[i] <= k[i - 1]) *flag = 0;   // benign race: every writer stores 0
}

extern "C" int pcc_keys_is_canonical(const int64_t* keys, int64_t n, int32_t* d_flag, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(d_flag, "pcc_keys_is_canonical: d_flag is NULL");
  PCC_CHECK_HIP(hipMemsetAsync(d_flag, 1, sizeof(int), s));   // non-zero = canonical
  if (n <= 1) return PCC_OK;
  k_is_canonical<<<grid1(n), 256, 0, s>>>(keys, n, d_flag);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// stride and generative expansion
// ------------------------------------------------------------------------------------------
__global__ void k_mask_keys(const int64_t* __restrict__ in, int64_t n, int64_t mask, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] & mask;
}

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" size_t pcc_stride_ws_bytes(int64_t n) {
  if (n <= 0) return 256;
  return 2 * pcc_align_up((size_t)n * 8) + pcc_sort_ws_bytes(n) + pcc_unique_ws_bytes(n) + 256;
}

extern "C" int pcc_coords_stride(const int64_t* keys, int64_t n, int32_t new_stride, uint64_t bit_mask,
                                 int64_t* out_keys, int64_t* d_count, void* ws, size_t ws_bytes,
                                 void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(d_count, "pcc_coords_stride: d_count is NULL");
  PCC_REQUIRE(is_pow2(new_stride) && new_stride <= (1 << 14), "pcc_coords_stride: tensor stride %d is not a power of two",
              new_stride);
  if (n <= 0) {
    PCC_CHECK_HIP(hipMemsetAsync(d_count, 0, sizeof(int64_t), s));
    return PCC_OK;
  }
  if (ws_bytes < pcc_stride_ws_bytes(n)) {
    pcc_set_error("pcc_coords_stride: workspace too small");
    return PCC_EWS;
  }
  // floor(c/m)*m on a biased field == clearing its low bits (2^15 is a multiple of m)
  const int64_t fm = 0xFFFF & ~(int64_t)(new_stride - 1);
  const int64_t mask = (int64_t)((0xFFFFull << 48) | ((uint64_t)fm << 32) | ((uint64_t)fm << 16) | (uint64_t)fm);
  char* p = (char*)ws;
  int64_t* masked = (int64_t*)p;  p += pcc_align_up((size_t)n * 8);
  int64_t* sorted = (int64_t*)p;  p += pcc_align_up((size_t)n * 8);
  void* sort_ws = p;              p += pcc_sort_ws_bytes(n);
  void* uniq_ws = p;
  k_mask_keys<<<grid1(n), 256, 0, s>>>(keys, n, mask, masked);
  PCC_LAUNCH_CHECK();
  PCC_TRY(pcc_sort_keys(masked, n, bit_mask & (uint64_t)mask, sorted, nullptr, sort_ws, pcc_sort_ws_bytes(n), s));
  PCC_TRY(pcc_unique_sorted(sorted, n, out_keys, nullptr, d_count, uniq_ws, pcc_unique_ws_bytes(n), s));
  return PCC_OK;
}

__global__ void k_expand(const int64_t* __restrict__ in, int64_t n, int K, int ks, int step,
                         int64_t* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * K) return;
  const int64_t i = t / K;
  const int k = (int)(t - i * K);
  out[t] = in[i] + pcc_delta_of(k, ks, step);
}

extern "C" size_t pcc_expand_ws_bytes(int64_t n, int32_t kernel_size) {
  if (n <= 0) return 256;
  const int64_t m = n * kernel_size * kernel_size * kernel_size;
  return 2 * pcc_align_up((size_t)m * 8) + pcc_sort_ws_bytes(m) + pcc_unique_ws_bytes(m) + 256;
}

extern "C" int pcc_coords_expand(const int64_t* keys, int64_t n, int32_t kernel_size, int32_t ts_out,
                                 uint64_t bit_mask, int64_t* out_keys, int64_t* d_count, void* ws,
                                 size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(d_count, "pcc_coords_expand: d_count is NULL");
  PCC_REQUIRE(kernel_size >= 1 && kernel_size <= 5 && ts_out >= 1, "pcc_coords_expand: unsupported kernel_size %d", kernel_size);
  if (n <= 0) {
    PCC_CHECK_HIP(hipMemsetAsync(d_count, 0, sizeof(int64_t), s));
    return PCC_OK;
  }
  const int K = kernel_size * kernel_size * kernel_size;
  const int64_t m = n * K;
  PCC_REQUIRE(m < (1ll << 31), "pcc_coords_expand: too many candidates");
  if (ws_bytes < pcc_expand_ws_bytes(n, kernel_size)) {
    pcc_set_error("pcc_coords_expand: workspace too small");
    return PCC_EWS;
  }
  char* p = (char*)ws;
  int64_t* cand = (int64_t*)p;    p += pcc_align_up((size_t)m * 8);
  int64_t* sorted = (int64_t*)p;  p += pcc_align_up((size_t)m * 8);
  void* sort_ws = p;              p += pcc_sort_ws_bytes(m);
  void* uniq_ws = p;
  k_expand<<<grid1(m), 256, 0, s>>>(keys, n, K, kernel_size, ts_out, cand);
  PCC_LAUNCH_CHECK();
  PCC_TRY(pcc_sort_keys(cand, m, bit_mask, sorted, nullptr, sort_ws, pcc_sort_ws_bytes(m), s));
  PCC_TRY(pcc_unique_sorted(sorted, m, out_keys, nullptr, d_count, uniq_ws, pcc_unique_ws_bytes(m), s));
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// Generative expansion fused with its transposed kernel map.
//   candidate t = i*K + k  ("pair id": input row i, kernel offset k)  ->  32-bit linear cell index of
//   key[i] + off_k in the OUTPUT lattice (order preserving: ascending cell == ascending canonical key).
//   One radix sort of (cell, pair id) + adjacent-unique yields the output coordinate set AND, for every output
//   row o, the contiguous list pair_ids[first[o] .. first[o+1]) of the pairs that land on it -- the whole
//   transposed map, with no neighbour search at all.  The sort is stable, so each list is ordered by pair id:
//   a fixed summation order (deterministic).
// ------------------------------------------------------------------------------------------
struct Lattice { int lo[3]; int dims[3]; int ts_log2; int nbatch; };

__global__ void k_expand_cells(const int64_t* __restrict__ in, int64_t n, int K, int ks, int step, Lattice L,
                               unsigned* __restrict__ cells) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * K) return;
  const int64_t i = t / K;
  const int k = (int)(t - i * K);
  const int64_t key = in[i] + pcc_delta_of(k, ks, step);
  const int b = (int)(key >> 48);
  const int cx = ((int)((key >> 32) & 0xFFFF) - (int)PCC_BIAS - L.lo[0]) >> L.ts_log2;
  const int cy = ((int)((key >> 16) & 0xFFFF) - (int)PCC_BIAS - L.lo[1]) >> L.ts_log2;
  const int cz = ((int)(key & 0xFFFF) - (int)PCC_BIAS - L.lo[2]) >> L.ts_log2;
  cells[t] = (unsigned)((((long long)b * L.dims[0] + cx) * L.dims[1] + cy) * L.dims[2] + cz);
}

__global__ void k_cell_flags(const unsigned* __restrict__ c, int64_t n, int* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || c[i] != c[i - 1]) ? 1 : 0;
}

__global__ void k_cell_unique(const unsigned* __restrict__ c, int64_t n, const int* __restrict__ pos, Lattice L,
                              int64_t* __restrict__ out_keys, int* __restrict__ first, int64_t* __restrict__ d_count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool f = (i == 0 || c[i] != c[i - 1]);
  if (f) {
    unsigned cell = c[i];
    const int cz = cell % L.dims[2]; cell /= L.dims[2];
    const int cy = cell % L.dims[1]; cell /= L.dims[1];
    const int cx = cell % L.dims[0]; cell /= L.dims[0];
    const int64_t x = ((int64_t)cx << L.ts_log2) + L.lo[0] + PCC_BIAS;
    const int64_t y = ((int64_t)cy << L.ts_log2) + L.lo[1] + PCC_BIAS;
    const int64_t z = ((int64_t)cz << L.ts_log2) + L.lo[2] + PCC_BIAS;
    out_keys[pos[i]] = ((int64_t)cell << 48) | (x << 32) | (y << 16) | z;
    first[pos[i]] = (int)i;
  }
  if (i == n - 1) {
    const int64_t cnt = (int64_t)pos[i] + (f ? 1 : 0);
    *d_count = cnt;
    first[cnt] = (int)n;     // CSR end sentinel
  }
}

extern "C" size_t pcc_expand_csr_ws_bytes(int64_t n, int32_t kernel_size) {
  if (n <= 0) return 256;
  const int64_t m = n * kernel_size * kernel_size * kernel_size;
  const int64_t nb = pcc_cdiv(m, RS_B);
  return 2 * pcc_align_up((size_t)m * 4) + pcc_align_up((size_t)m * 4) + pcc_align_up((size_t)nb * 256 * 4) +
         pcc_scan_ws_bytes(nb * 256) + pcc_scan_ws_bytes(m) + 1024;
}

extern "C" int pcc_coords_expand_csr(const int64_t* keys, int64_t n, int32_t kernel_size, int32_t ts_out,
                                     const int32_t* h_lattice, int64_t* out_keys, int64_t* d_count,
                                     int32_t* pair_ids, int32_t* first, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(d_count && h_lattice, "pcc_coords_expand_csr: NULL argument");
  PCC_REQUIRE(kernel_size >= 1 && kernel_size <= 5 && ts_out >= 1 && (ts_out & (ts_out - 1)) == 0,
              "pcc_coords_expand_csr: unsupported kernel_size %d / ts_out %d", kernel_size, ts_out);
  if (n <= 0) {
    PCC_CHECK_HIP(hipMemsetAsync(d_count, 0, sizeof(int64_t), s));
    if (first) PCC_CHECK_HIP(hipMemsetAsync(first, 0, sizeof(int32_t), s));
    return PCC_OK;
  }
  PCC_REQUIRE(keys && out_keys && pair_ids && first && ws, "pcc_coords_expand_csr: NULL array");
  Lattice L;
  for (int i = 0; i < 3; ++i) { L.lo[i] = h_lattice[i]; L.dims[i] = h_lattice[3 + i]; }
  PCC_REQUIRE(h_lattice[6] == ts_out, "pcc_coords_expand_csr: lattice pitch must equal ts_out");
  int l = 0; while ((1 << l) < ts_out) ++l;
  L.ts_log2 = l; L.nbatch = h_lattice[7];
  const long long cells = (long long)L.nbatch * L.dims[0] * L.dims[1] * L.dims[2];
  PCC_REQUIRE(cells > 0 && cells <= 0xFFFFFFFFll, "pcc_coords_expand_csr: lattice has %lld cells (needs <= 2^32; use pcc_coords_expand)", cells);
  const int K = kernel_size * kernel_size * kernel_size;
  const int64_t m = n * K;
  PCC_REQUIRE(m < (1ll << 31), "pcc_coords_expand_csr: too many candidates");
  if (ws_bytes < pcc_expand_csr_ws_bytes(n, kernel_size)) {
    pcc_set_error("pcc_coords_expand_csr: workspace too small");
    return PCC_EWS;
  }
  const int64_t nb = pcc_cdiv(m, RS_B);
  char* p = (char*)ws;
  unsigned* ka = (unsigned*)p;  p += pcc_align_up((size_t)m * 4);
  unsigned* kb = (unsigned*)p;  p += pcc_align_up((size_t)m * 4);
  int* pb = (int*)p;            p += pcc_align_up((size_t)m * 4);     // payload ping-pong partner of pair_ids
  int* hist = (int*)p;          p += pcc_align_up((size_t)nb * 256 * 4);
  void* scan_ws = p;
  const size_t scan_bytes = ws_bytes - (size_t)(p - (char*)ws);
  k_expand_cells<<<grid1(m), 256, 0, s>>>(keys, n, K, kernel_size, ts_out, L, ka);
  PCC_LAUNCH_CHECK();
  int np = 0;
  while (np < 4 && (cells - 1) >> (8 * np)) ++np;     // 8-bit digits that can differ
  if (np == 0) np = 1;
  // payload must end in pair_ids: with np passes the first destination alternates accordingly
  const unsigned* src_k = ka;
  const int* src_p = nullptr;
  for (int i = 0; i < np; ++i) {
    const bool to_out = ((np - 1 - i) % 2) == 0;
    unsigned* dst_k = (src_k == ka) ? kb : ka;
    int* dst_p = to_out ? pair_ids : pb;
    k_rs_hist<unsigned><<<(unsigned)nb, RS_T, 0, s>>>(src_k, m, 8 * i, (int)nb, hist);
    PCC_LAUNCH_CHECK();
    PCC_TRY(pcc_scan_exclusive_i32(hist, hist, nb * 256, scan_ws, scan_bytes, s));
    k_rs_scatter<unsigned, true><<<(unsigned)nb, RS_T, 0, s>>>(src_k, src_p, m, 8 * i, (int)nb, hist, dst_k, dst_p);
    PCC_LAUNCH_CHECK();
    src_k = dst_k;
    src_p = dst_p;
  }
  // unique over the sorted cells; `pos` re-uses the histogram scratch region sized for m ints
  int* pos = (int*)((src_k == ka) ? kb : ka);          // the idle key buffer holds m ints
  k_cell_flags<<<grid1(m), 256, 0, s>>>(src_k, m, pos);
  PCC_LAUNCH_CHECK();
  PCC_TRY(pcc_scan_exclusive_i32(pos, pos, m, scan_ws, scan_bytes, s));
  k_cell_unique<<<grid1(m), 256, 0, s>>>(src_k, m, pos, L, out_keys, first, d_count);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// Bounds of user coordinates (what `Bounds` / the grid lattices are sized from) and the canonical-order row gather
// of user-ordered features: the two torch reductions / gathers that were left on the inference path in round 1.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_coords_bounds(const T* __restrict__ coords, long long n, int* __restrict__ out8) {
  __shared__ int s_min[4], s_max[4];
  if (threadIdx.x < 4) { s_min[threadIdx.x] = 0x7FFFFFFF; s_max[threadIdx.x] = (int)0x80000000; }
  __syncthreads();
  int mn[4] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, mx[4] = {(int)0x80000000, (int)0x80000000, (int)0x80000000, (int)0x80000000};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int v = (int)floorf((float)coords[i * 4 + c]);
      mn[c] = min(mn[c], v); mx[c] = max(mx[c], v);
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    for (int d = 32; d >= 1; d >>= 1) { mn[c] = min(mn[c], __shfl_xor(mn[c], d)); mx[c] = max(mx[c], __shfl_xor(mx[c], d)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&s_min[c], mn[c]); atomicMax(&s_max[c], mx[c]); }
  }
  __syncthreads();
  if (threadIdx.x < 4) { atomicMin(&out8[threadIdx.x], s_min[threadIdx.x]); atomicMax(&out8[4 + threadIdx.x], s_max[threadIdx.x]); }
}

// out8 = {min b, min x, min y, min z, max b, max x, max y, max z} (floor of float coordinates); n >= 1
extern "C" int pcc_coords_bounds(const void* coords, int32_t is_float, int64_t n, int32_t* out8, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(coords && out8 && n >= 1, "pcc_coords_bounds: bad arguments");
  PCC_CHECK_HIP(hipMemsetAsync(out8, 0x7F, 16, s));
  PCC_CHECK_HIP(hipMemsetAsync(out8 + 4, 0x80, 16, s));
  const unsigned g = (unsigned)(pcc_cdiv(n, 256) < 1024 ? pcc_cdiv(n, 256) : 1024);
  if (is_float) k_coords_bounds<float><<<g, 256, 0, s>>>((const float*)coords, n, out8);
  else k_coords_bounds<int><<<g, 256, 0, s>>>((const int*)coords, n, out8);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ---- one-kernel intake of a frame and one-kernel hand-over of a decoded block ----------------------------------------------
// `UnifiedModel.compress` turns the [n, 6] frame (x y z r g b) into coordinates [0, x, y, z] and features [1, r, g, b]
// (reference `model/model.py:141-161`); the constructor of its sparse tensor then needs the bounds of the floored coordinates
// and whether the rows already are in canonical order.  Ten element-wise launches (zeros, two concatenations, a cast, the key
// packing, two fills, the bounds, the order check) in the host-bound opening of a step: one pass here.
// out12: [0..3] min (b, x, y, z), [4..7] MINUS max (everything is a minimum), [8] != 0: canonical order.
// partial row of a workgroup: [0..3] min (b, x, y, z), [4..7] min of MINUS (b, x, y, z), [8] 1 unless some row is not above
// its predecessor; 16 ints per workgroup, reduced by k_intake_reduce (no global atomics: 3 000 same-line atomics of the first
// version cost 60 us)
__device__ __forceinline__ void intake_block_reduce(int (&v)[9], int* __restrict__ part) {
  __shared__ int s_red[4][9];
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    for (int d = 32; d >= 1; d >>= 1) v[c] = min(v[c], __shfl_xor(v[c], d));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][c] = v[c];
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int c = threadIdx.x;
    part[blockIdx.x * 16 + c] = min(min(s_red[0][c], s_red[1][c]), min(s_red[2][c], s_red[3][c]));
  }
}

__global__ void __launch_bounds__(256) k_frame_intake(const float* __restrict__ pc, long long n, long long* __restrict__ keys,
                                                      float4* __restrict__ feats, int* __restrict__ part) {
  int v[9] = {0, 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, 0, 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, 1};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float2* r = reinterpret_cast<const float2*>(pc + i * 6);
    const float2 a = r[0], b = r[1], c = r[2];                       // x y | z r | g b
    const int x = (int)floorf(a.x), y = (int)floorf(a.y), z = (int)floorf(b.x);
    const long long k = pack4(0, x, y, z);
    keys[i] = k;
    feats[i] = make_float4(1.f, b.y, c.x, c.y);
    v[1] = min(v[1], x); v[2] = min(v[2], y); v[3] = min(v[3], z);
    v[5] = min(v[5], -x); v[6] = min(v[6], -y); v[7] = min(v[7], -z);
    if (i > 0) {
      const float* q = pc + (i - 1) * 6;
      if (k <= pack4(0, (int)floorf(q[0]), (int)floorf(q[1]), (int)floorf(q[2]))) v[8] = 0;
    }
  }
  intake_block_reduce(v, part);
}

// the same for an int32 [n, 4] coordinate tensor (b, x, y, z): what `decompress` receives for the latent
__global__ void __launch_bounds__(256) k_coords_intake_i32(const int4* __restrict__ c4, long long n, long long* __restrict__ keys,
                                                           int* __restrict__ part) {
  int v[9] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, 1};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int4 c = c4[i];
    const long long k = pack4(c.x, c.y, c.z, c.w);
    keys[i] = k;
    v[0] = min(v[0], c.x); v[1] = min(v[1], c.y); v[2] = min(v[2], c.z); v[3] = min(v[3], c.w);
    v[4] = min(v[4], -c.x); v[5] = min(v[5], -c.y); v[6] = min(v[6], -c.z); v[7] = min(v[7], -c.w);
    if (i > 0) {
      const int4 q = c4[i - 1];
      if (k <= pack4(q.x, q.y, q.z, q.w)) v[8] = 0;
    }
  }
  intake_block_reduce(v, part);
}

__global__ void __launch_bounds__(256) k_intake_reduce(const int* __restrict__ part, int nblk, int* __restrict__ out12) {
  __shared__ int s_red[4][9];
  int v[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) v[c] = 0x7FFFFFFF;
  for (int b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
    for (int c = 0; c < 9; ++c) v[c] = min(v[c], part[b * 16 + c]);
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    for (int d = 32; d >= 1; d >>= 1) v[c] = min(v[c], __shfl_xor(v[c], d));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][c] = v[c];
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int c = threadIdx.x;
    out12[c] = min(min(s_red[0][c], s_red[1][c]), min(s_red[2][c], s_red[3][c]));
  }
}

extern "C" size_t pcc_frame_intake_ws_bytes(void) { return (size_t)1024 * 16 * sizeof(int); }

extern "C" int pcc_frame_intake(const float* pc, int64_t n, int64_t* keys, float* feats, int32_t* out12, void* ws, size_t ws_bytes,
                                void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(pc && keys && feats && out12 && ws && n >= 1, "pcc_frame_intake: bad arguments");
  PCC_REQUIRE((((uintptr_t)pc & 7) | ((uintptr_t)feats & 15)) == 0, "pcc_frame_intake: the frame must be 8-byte, the features 16-byte aligned");
  if (ws_bytes < pcc_frame_intake_ws_bytes()) { pcc_set_error("pcc_frame_intake: workspace too small"); return PCC_EWS; }
  const unsigned g = (unsigned)(pcc_cdiv(n, 256) < 1024 ? pcc_cdiv(n, 256) : 1024);
  k_frame_intake<<<g, 256, 0, s>>>(pc, n, (long long*)keys, (float4*)feats, (int*)ws);
  k_intake_reduce<<<1, 256, 0, s>>>((const int*)ws, (int)g, out12);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_coords_intake_i32(const int32_t* coords, int64_t n, int64_t* keys, int32_t* out12, void* ws, size_t ws_bytes,
                                     void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(coords && keys && out12 && ws && n >= 1 && ((uintptr_t)coords & 15) == 0, "pcc_coords_intake_i32: bad arguments");
  if (ws_bytes < pcc_frame_intake_ws_bytes()) { pcc_set_error("pcc_coords_intake_i32: workspace too small"); return PCC_EWS; }
  const unsigned g = (unsigned)(pcc_cdiv(n, 256) < 1024 ? pcc_cdiv(n, 256) : 1024);
  k_coords_intake_i32<<<g, 256, 0, s>>>((const int4*)coords, n, (long long*)keys, (int*)ws);
  k_intake_reduce<<<1, 256, 0, s>>>((const int*)ws, (int)g, out12);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// `UnifiedModel.decompress` returns [x, y, z, clamp(round(255 f), 0, 255) / 255] (reference `model/model.py:240-250`): the
// coordinates straight from the keys of the decoded set, the colours in the same fp32 operations as the torch chain
// (round half to even; NaN stays NaN), one launch instead of nine.
__global__ void __launch_bounds__(256) k_decode_finish(const long long* __restrict__ keys, const float* __restrict__ f, long long n,
                                                       float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long k = keys[i];
  float v[6];
  v[0] = (float)((int)((k >> 32) & 0xFFFF) - (int)PCC_BIAS);
  v[1] = (float)((int)((k >> 16) & 0xFFFF) - (int)PCC_BIAS);
  v[2] = (float)((int)(k & 0xFFFF) - (int)PCC_BIAS);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float t = rintf(f[i * 3 + c] * 255.0f);
    t = t < 0.0f ? 0.0f : t;
    t = t > 255.0f ? 255.0f : t;
    v[3 + c] = t * (1.0f / 255.0f);                     // (torch divides by a host scalar as a multiplication by its fp32 reciprocal)
  }
  float2* o = reinterpret_cast<float2*>(out + i * 6);
  o[0] = make_float2(v[0], v[1]); o[1] = make_float2(v[2], v[3]); o[2] = make_float2(v[4], v[5]);
}

extern "C" int pcc_decode_finish(const int64_t* keys, const float* feats3, int64_t n, float* out6, void* stream) {
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(keys && feats3 && out6 && ((uintptr_t)out6 & 7) == 0, "pcc_decode_finish: bad arguments");
  k_decode_finish<<<grid1(n), 256, 0, (hipStream_t)stream>>>((const long long*)keys, feats3, n, out6);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

__global__ void __launch_bounds__(256) k_rows_gather(const float* __restrict__ src, const long long* __restrict__ idx, long long m,
                                                     int c, float* __restrict__ dst) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= m * c) return;
  const long long r = t / c;
  dst[t] = src[idx[r] * c + (t - r * c)];
}

// dst[i][:] = src[idx[i]][:]  (SparseTensor: features of user-ordered rows in canonical order)
extern "C" int pcc_rows_gather(const float* src, const int64_t* idx, int64_t m, int32_t c, float* dst, void* stream) {
  if (m <= 0) return PCC_OK;
  PCC_REQUIRE(src && idx && dst && c >= 1, "pcc_rows_gather: bad arguments");
  k_rows_gather<<<(unsigned)pcc_cdiv(m * c, 256), 256, 0, (hipStream_t)stream>>>(src, (const long long*)idx, m, c, dst);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}
