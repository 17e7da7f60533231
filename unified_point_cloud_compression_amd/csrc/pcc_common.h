// Shared helpers of libpcc_hip.so (gfx950 only; wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "pcc_hip.h"

void pcc_set_error(const char* fmt, ...);

#define PCC_CHECK_HIP(expr)                                                                    \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      pcc_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_));       \
      return PCC_EHIP;                                                                         \
    }                                                                                          \
  } while (0)

#define PCC_REQUIRE(cond, ...)                                                                 \
  do {                                                                                         \
    if (!(cond)) {                                                                             \
      pcc_set_error(__VA_ARGS__);                                                              \
      return PCC_EINVAL;                                                                       \
    }                                                                                          \
  } while (0)

#define PCC_LAUNCH_CHECK() PCC_CHECK_HIP(hipGetLastError())

#define PCC_TRY(expr)                                                                          \
  do {                                                                                         \
    int rc_ = (expr);                                                                          \
    if (rc_ != PCC_OK) return rc_;                                                             \
  } while (0)

static constexpr int PCC_WAVE = 64;
static constexpr int64_t PCC_BIAS = 1 << 15;

// ---- map header layout (int32 words), see include/pcc_hip.h ---------------------------------
static constexpr int HDR_NSEG = 0;
static constexpr int HDR_K = 1;
static constexpr int HDR_FLAGS = 2;        // bit0: offsets of segment are listed in koffs[]
static constexpr int HDR_SEG0 = 8;         // 6 words per segment
static constexpr int SEG_POS_BEGIN = 0, SEG_POS_COUNT = 1, SEG_K_COUNT = 2, SEG_KOFF_BEGIN = 3,
                     SEG_NBR_LO = 4, SEG_NBR_HI = 5;
static constexpr int SEG_WORDS = 6;
static constexpr int HDR_KOFFS = 64;       // int32[K] kernel-offset id of each listed offset
static_assert(PCC_MAP_HDR_INTS >= HDR_KOFFS + 192, "header too small");

static inline size_t pcc_align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int64_t pcc_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// kernel offset id -> (dx,dy,dz); x fastest.  odd k: centred, even k: 0..k-1.
__host__ __device__ inline void pcc_offset_of(int kid, int ks, int& dx, int& dy, int& dz) {
  const int lo = (ks & 1) ? -(ks - 1) / 2 : 0;
  dx = kid % ks + lo;
  dy = (kid / ks) % ks + lo;
  dz = kid / (ks * ks) + lo;
}

__host__ __device__ inline int64_t pcc_delta_of(int kid, int ks, int step) {
  int dx, dy, dz;
  pcc_offset_of(kid, ks, dx, dy, dz);
  return (int64_t)dx * step * (1ll << 32) + (int64_t)dy * step * (1ll << 16) + (int64_t)dz * step;
}

// binary search in an ascending key array; -1 when absent
__device__ inline int pcc_find(const int64_t* __restrict__ keys, int n, int64_t q) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < q) lo = mid + 1; else hi = mid;
  }
  return (lo < n && keys[lo] == q) ? lo : -1;
}

// Wave-cooperative search: the 64 queries of a wave are usually close together (consecutive sorted rows shifted by
// one offset), so the wave first brackets [lower_bound(min q), upper_bound(max q)) with wave-uniform probes (scalar
// loads through the constant cache), then every lane searches only inside that short bracket.  Correct for any
// query distribution; all 64 lanes must call it (inactive lanes pass a key of an active one).
__device__ inline int64_t pcc_wave_uniform(int64_t v) {
  const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xFFFFFFFFll));
  const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((int64_t)hi << 32) | (int64_t)(unsigned)lo;
}

__device__ inline int pcc_find_bracketed(const int64_t* __restrict__ keys, int n, int64_t q) {
  int64_t qmin = q, qmax = q;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int64_t a = __shfl_xor(qmin, d), b = __shfl_xor(qmax, d);
    qmin = a < qmin ? a : qmin;
    qmax = b > qmax ? b : qmax;
  }
  qmin = pcc_wave_uniform(qmin);
  qmax = pcc_wave_uniform(qmax);
  int l = 0, h = n;
  while (l < h) { const int m = (l + h) >> 1; if (keys[m] < qmin) l = m + 1; else h = m; }
  int lo = l;
  h = n;
  while (l < h) { const int m = (l + h) >> 1; if (keys[m] <= qmax) l = m + 1; else h = m; }
  int hi = l;
  const int end = hi;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < q) lo = mid + 1; else hi = mid;
  }
  return (lo < end && keys[lo] == q) ? lo : -1;
}

// internal cross-file entry points
int pcc_scan_exclusive_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes,
                           hipStream_t s);
size_t pcc_scan_ws_bytes(int64_t n);
