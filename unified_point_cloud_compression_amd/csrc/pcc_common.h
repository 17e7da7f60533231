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
static constexpr int HDR_ORDER = 256;      // int32[K] per segment: slot visiting order (z innermost, see pcc_map.hip)
static_assert(PCC_MAP_HDR_INTS >= HDR_ORDER + 192, "header too small");

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

// ---- grid index: occupancy bitmap + rank over the bounding lattice of a canonical set -----------------
// Canonical order (b,x,y,z ascending) equals ascending linear cell index ((b*nx+cx)*ny+cy)*nz+cz, so the row of
// an occupied cell is rank[word] + popcount(bits below it): two coalesced reads instead of a 20-step search.
struct PccGrid {
  const unsigned long long* bits;   // nullptr: no grid, fall back to binary search
  const int* rank;                  // exclusive prefix popcount per 64-bit word
  int lo[3];                        // coordinate of cell 0 per axis
  int dims[3];                      // cells per axis
  int ts_log2;                      // lattice pitch = 1 << ts_log2
  int nbatch;
};

__device__ inline int pcc_grid_find(const PccGrid& g, int64_t key) {
  const int b = (int)(key >> 48);
  const int x = (int)((key >> 32) & 0xFFFF) - (int)PCC_BIAS - g.lo[0];
  const int y = (int)((key >> 16) & 0xFFFF) - (int)PCC_BIAS - g.lo[1];
  const int z = (int)(key & 0xFFFF) - (int)PCC_BIAS - g.lo[2];
  if ((x | y | z) < 0 || b >= g.nbatch) return -1;
  if ((x | y | z) & ((1 << g.ts_log2) - 1)) return -1;   // off-lattice query (inverse maps of strided convs ask these)
  const int cx = x >> g.ts_log2, cy = y >> g.ts_log2, cz = z >> g.ts_log2;
  if (cx >= g.dims[0] || cy >= g.dims[1] || cz >= g.dims[2]) return -1;
  const long long cell = (((long long)b * g.dims[0] + cx) * g.dims[1] + cy) * g.dims[2] + cz;
  const unsigned long long w = g.bits[cell >> 6];
  const int bit = (int)(cell & 63);
  if (!((w >> bit) & 1ull)) return -1;
  return g.rank[cell >> 6] + __popcll(w & ((1ull << bit) - 1ull));
}

__device__ inline int pcc_lookup(const PccGrid& g, const int64_t* __restrict__ keys, int n, int64_t q) {
  return g.bits ? pcc_grid_find(g, q) : pcc_find(keys, n, q);
}

// The 3x3x3 neighbourhood of a row of a set straight from the set's own grid index (pitch = the grid's pitch): the z cells
// of a (dx,dy) column are one bit field of the occupancy bitmap, the row of its first occupied cell is rank + popcount and
// the others follow consecutively.  Columns c = (dx+1) + 3*(dy+1) with c % nlanes == lane are handled (nlanes lanes share a
// row); returns the presence mask of those columns (bit k = neighbour k exists, k = (dx+1) + 3(dy+1) + 9(dz+1), the kernel
// offset enumeration) and, when rows != nullptr, rows[k] for the present ones.
__device__ inline unsigned pcc_grid_nbr27(const PccGrid& g, int64_t key, int lane, int nlanes, int* rows) {
  const int b = (int)(key >> 48);
  const int cx = (((int)((key >> 32) & 0xFFFF) - (int)PCC_BIAS - g.lo[0]) >> g.ts_log2);
  const int cy = (((int)((key >> 16) & 0xFFFF) - (int)PCC_BIAS - g.lo[1]) >> g.ts_log2);
  const int cz = (((int)(key & 0xFFFF) - (int)PCC_BIAS - g.lo[2]) >> g.ts_log2);
  const int z_lo = cz > 0 ? cz - 1 : 0, z_hi = cz + 1 < g.dims[2] ? cz + 1 : g.dims[2] - 1;
  const int nz = z_hi - z_lo + 1;
  unsigned mask = 0;
  for (int c = lane; c < 9; c += nlanes) {
    const int nx = cx + c % 3 - 1, ny = cy + c / 3 - 1;
    if (nx < 0 || ny < 0 || nx >= g.dims[0] || ny >= g.dims[1]) continue;
    const long long cell = (((long long)b * g.dims[0] + nx) * g.dims[1] + ny) * g.dims[2] + z_lo;
    const long long wi = cell >> 6;
    const int sh = (int)(cell & 63);
    const unsigned long long w0 = g.bits[wi];
    unsigned long long f64 = w0 >> sh;
    if (sh + nz > 64) f64 |= g.bits[wi + 1] << (64 - sh);
    unsigned f = (unsigned)f64 & ((1u << nz) - 1u);
    if (!f) continue;
    if (!rows) {                         // presence only: the field's bits go to k = c + 9 * (first dz + t), t = 0..2
      mask |= ((f & 1u) | ((f & 2u) << 8) | ((f & 4u) << 16)) << (c + 9 * (z_lo - cz + 1));
      continue;
    }
    int r = g.rank[wi] + __popcll(w0 & ((1ull << sh) - 1ull));
    while (f) {
      const int t = __ffs((int)f) - 1;
      f &= f - 1;
      const int k = c + 9 * (z_lo + t - cz + 1);
      mask |= 1u << k;
      rows[k] = r++;
    }
  }
  return mask;
}

// internal cross-file entry points
int pcc_scan_exclusive_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes,
                           hipStream_t s);
size_t pcc_scan_ws_bytes(int64_t n);
