// Kernel-map build over canonical (sorted) coordinate keys.
//
// conv map:        one segment, nbr[k][o] = row of (out_key[o] + off_k*step) in in_keys.
// transposed map:  output rows are grouped (stably) by their residue class modulo the up-sampling
//                  stride; a class lists only the kernel offsets congruent to its residue, so no
//                  parity-impossible (in,out,k) candidate is ever stored or multiplied.
// Queries of consecutive rows for one offset are monotone, so the binary searches of a wave walk
// the same few cache lines of the sorted key array (L2-resident, coalesced).
#include <string.h>

#include "pcc_common.h"

struct SegPlan {           // host-built, passed by value to k_write_hdr
  int nseg;
  int K;
  int listed;              // 1: koffs[] lists the offsets of each segment
  int k_count[PCC_MAP_MAX_SEG];
  int koff_begin[PCC_MAP_MAX_SEG];
  unsigned char koffs[192];
  unsigned char order[192];  // per segment: visiting order of its slots (segment-local slot indices)
};

// class_begin: nullptr -> single segment covering [0, n_out)
// threads 0..191 of ONE workgroup: thread i writes the i-th offset / order entry, thread 0 the segment table (one thread walking
// all of it took 10-14 us per map, ten maps per step)
__device__ __forceinline__ void write_hdr(const SegPlan& plan, const int* __restrict__ class_begin, int class_stride,
                                          int64_t n_out, int* __restrict__ hdr) {
  {
    const int i = (int)threadIdx.x;
    if (i < plan.K && i < 192) {
      hdr[HDR_KOFFS + i] = plan.listed ? plan.koffs[i] : i;
      hdr[HDR_ORDER + i] = plan.order[i];
    }
  }
  if (threadIdx.x != 0) return;
  hdr[HDR_NSEG] = plan.nseg;
  hdr[HDR_K] = plan.K;
  hdr[HDR_FLAGS] = plan.listed;
  int64_t nbr_begin = 0;
  for (int s = 0; s < plan.nseg; ++s) {
    int pb = 0, pc = (int)n_out;
    if (class_begin) {
      pb = class_begin[s * class_stride];
      const int pe = (s + 1 < plan.nseg) ? class_begin[(s + 1) * class_stride] : (int)n_out;
      pc = pe - pb;
    }
    int* seg = hdr + HDR_SEG0 + s * SEG_WORDS;
    seg[SEG_POS_BEGIN] = pb;
    seg[SEG_POS_COUNT] = pc;
    seg[SEG_K_COUNT] = plan.k_count[s];
    seg[SEG_KOFF_BEGIN] = plan.koff_begin[s];
    seg[SEG_NBR_LO] = (int)(nbr_begin & 0xFFFFFFFFll);
    seg[SEG_NBR_HI] = (int)(nbr_begin >> 32);
    nbr_begin += (int64_t)pc * plan.k_count[s];
  }
}

__global__ void k_write_hdr(SegPlan plan, const int* __restrict__ class_begin, int class_stride,
                            int64_t n_out, int* __restrict__ hdr) {
  if (blockIdx.x != 0) return;                              // launched <<<1, 192>>>
  write_hdr(plan, class_begin, class_stride, n_out, hdr);
}

// Visiting order of a segment's offsets.  Tried on MI355X (round 1): "(dx,dy) major, dz minor" -- back-to-back
// visits of the three dz-neighbours, which are the same input rows shifted by one lane -- made the gather-bound
// kernels 20 % SLOWER (misses on lines still in flight serialise); the natural kernel-offset order stays.
static constexpr bool ORDER_Z_INNERMOST = false;
static void plan_order(SegPlan& plan, int ks) {
  for (int s = 0; s < plan.nseg; ++s) {
    const int kb = plan.koff_begin[s], kc = plan.k_count[s];
    int key[192];
    for (int j = 0; j < kc; ++j) {
      const int kid = plan.listed ? plan.koffs[kb + j] : (kb + j);
      key[j] = (kid % (ks * ks)) * ks + kid / (ks * ks);
      plan.order[kb + j] = (unsigned char)j;
    }
    if (!ORDER_Z_INNERMOST) continue;
    for (int a = 1; a < kc; ++a)            // insertion sort of <= 125 entries
      for (int b = a; b > 0 && key[plan.order[kb + b]] < key[plan.order[kb + b - 1]]; --b) {
        const unsigned char t = plan.order[kb + b]; plan.order[kb + b] = plan.order[kb + b - 1]; plan.order[kb + b - 1] = t;
      }
  }
}

// pair counting without global atomics: one partial per block, summed by k_sum_counts
__device__ inline void count_pairs(bool hit, int* __restrict__ block_counts) {
  if (!block_counts) return;   // kernel-uniform
  __shared__ int wc[4];
  const unsigned long long m = __ballot(hit);
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0)
    block_counts[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

__global__ void __launch_bounds__(1024) k_sum_counts(const int* __restrict__ c, int64_t n,
                                                     int64_t* __restrict__ total) {
  __shared__ long long ws[16];
  long long s = 0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += c[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long t = 0;
    for (int i = 0; i < 16; ++i) t += ws[i];
    *total = t;
  }
}

// conv: thread per (k, o), o fastest
__global__ void __launch_bounds__(256) k_map_conv(PccGrid grid, const int64_t* __restrict__ in_keys, int n_in,
                                                  const int64_t* __restrict__ out_keys, int64_t n_out, int ks,
                                                  int step, const int* __restrict__ rows, int* __restrict__ nbr,
                                                  int* __restrict__ d_pairs) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // position
  const int k = blockIdx.y;
  int idx = -1;
  if (o < n_out) {
    const int64_t key = out_keys[rows ? rows[o] : o];
    idx = pcc_lookup(grid, in_keys, n_in, key + pcc_delta_of(k, ks, step));
    nbr[(int64_t)k * n_out + o] = idx;
  }
  count_pairs(idx >= 0, d_pairs);
}

// conv through the grid index: thread per (kernel column (kx, ky), o) walks the KS offsets along z.  z is the fastest
// cell axis, so the KS probes of a thread fall in one or two bitmap words: one key read, one (rarely two) word + rank
// reads for KS neighbours, instead of a key, a word and a rank per neighbour.
template <int KS>
__global__ void __launch_bounds__(256) k_map_conv_z(PccGrid g, const int64_t* __restrict__ out_keys, int64_t n_out,
                                                    int step, const int* __restrict__ rows, int* __restrict__ nbr,
                                                    int* __restrict__ d_pairs, SegPlan plan, int* __restrict__ hdr) {
  // (the map's header rides along: nothing in this kernel reads it, and a launch of its own cost ~5 us, nine times per step)
  if (hdr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 192) write_hdr(plan, nullptr, 0, n_out, hdr);
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int kxy = blockIdx.y;                                      // kx + KS * ky
  constexpr int H = (KS & 1) ? (KS - 1) / 2 : 0;
  int hits = 0;
  if (o < n_out) {
    const int64_t key = out_keys[rows ? rows[o] : o];
    const int b = (int)(key >> 48);
    const int x = (int)((key >> 32) & 0xFFFF) - (int)PCC_BIAS - g.lo[0] + (kxy % KS - H) * step;
    const int y = (int)((key >> 16) & 0xFFFF) - (int)PCC_BIAS - g.lo[1] + (kxy / KS - H) * step;
    const int z0 = (int)(key & 0xFFFF) - (int)PCC_BIAS - g.lo[2];
    const int tm = (1 << g.ts_log2) - 1;
    const int cx = x >> g.ts_log2, cy = y >> g.ts_log2;
    const bool col_ok = (x | y) >= 0 && b < g.nbatch && !((x | y) & tm) && cx < g.dims[0] && cy < g.dims[1];
    const long long cell0 = (((long long)b * g.dims[0] + cx) * g.dims[1] + cy) * g.dims[2];
    long long wi = -1;
    unsigned long long w = 0;
    int rk = 0;
#pragma unroll
    for (int iz = 0; iz < KS; ++iz) {
      const int z = z0 + (iz - H) * step;
      const int cz = z >> g.ts_log2;
      int idx = -1;
      if (col_ok && z >= 0 && !(z & tm) && cz < g.dims[2]) {
        const long long cell = cell0 + cz;
        if ((cell >> 6) != wi) { wi = cell >> 6; w = g.bits[wi]; rk = g.rank[wi]; }
        const int bit = (int)(cell & 63);
        if ((w >> bit) & 1ull) idx = rk + __popcll(w & ((1ull << bit) - 1ull));
      }
      nbr[(int64_t)(kxy + KS * KS * iz) * n_out + o] = idx;
      hits += idx >= 0;
    }
  }
  if (d_pairs) {                                                   // kernel-uniform
    __shared__ int wc[4];
    int c = hits;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) d_pairs[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
  }
}

// Morton (Z-curve) code of the lattice cell of every output row: bits of z, y, x interleaved (z lowest).  Tiles of
// consecutive positions in this order are compact 3-D blobs, so the 27 (or 125) neighbour gathers of a tile hit a
// few hundred distinct rows instead of a few thousand, and consecutive tiles share them (L1 / L2 locality).
__device__ inline uint64_t spread3(uint32_t v) {     // 16 bits -> every third bit
  uint64_t x = v & 0xFFFFull;
  x = (x | (x << 32)) & 0x00FF00000000FFFFull;
  x = (x | (x << 16)) & 0x00FF0000FF0000FFull;
  x = (x | (x << 8)) & 0xF00F00F00F00F00Full;
  x = (x | (x << 4)) & 0x30C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x9249249249249249ull;
  return x;
}

__global__ void k_morton(const int64_t* __restrict__ keys, int64_t n, int log2_step, uint64_t* __restrict__ code) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t k = keys[i];
  const uint32_t x = (uint32_t)((k >> 32) & 0xFFFF) >> log2_step;
  const uint32_t y = (uint32_t)((k >> 16) & 0xFFFF) >> log2_step;
  const uint32_t z = (uint32_t)(k & 0xFFFF) >> log2_step;
  const uint64_t b = (uint64_t)(k >> 48) & 0xFFFF;
  // batch index stays the major key (48 bits of interleaved coordinates below it); 16-bit batch ids would overflow
  // 64 bits together with 48 coordinate bits only for >= 2^16 batches, which the key format excludes
  code[i] = (b << 48) | spread3(z) | (spread3(y) << 1) | (spread3(x) << 2);
}

// transposed: class id per output row (as a sortable 64-bit key)
__global__ void k_classify(const int64_t* __restrict__ out_keys, int64_t n_out, int log2_step, int stride,
                           uint64_t* __restrict__ cls) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_out) return;
  const int64_t k = out_keys[o];
  const int m = stride - 1;
  const int rx = (int)((k >> 32) & 0xFFFF) >> log2_step & m;
  const int ry = (int)((k >> 16) & 0xFFFF) >> log2_step & m;
  const int rz = (int)(k & 0xFFFF) >> log2_step & m;
  cls[o] = (uint64_t)(rx + stride * (ry + stride * rz));
}

// class begin positions from the class-sorted id array (ids ascending): begin[c] = first p with id>=c
__global__ void k_class_begin(const uint64_t* __restrict__ sorted_cls, int64_t n, int nclass,
                              int* __restrict__ begin) {
  const int c = threadIdx.x;
  if (c >= nclass) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted_cls[mid] < (uint64_t)c) lo = mid + 1; else hi = mid;
  }
  begin[c] = (int)lo;
}

// transposed: thread per (slot j, position p), p fastest; in = out - off*step
__global__ void __launch_bounds__(256) k_map_transposed(PccGrid grid, const int64_t* __restrict__ in_keys, int n_in,
                                                        const int64_t* __restrict__ out_keys, int64_t n_out,
                                                        int ks, int step, const int* __restrict__ hdr,
                                                        const int* __restrict__ rows, int* __restrict__ nbr,
                                                        int* __restrict__ d_pairs) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  bool hit = false;
  if (p < n_out) {
    const int nseg = hdr[HDR_NSEG];
    int s = 0;
    for (; s < nseg - 1; ++s) {
      const int* sg = hdr + HDR_SEG0 + s * SEG_WORDS;
      if (p < (int64_t)sg[SEG_POS_BEGIN] + sg[SEG_POS_COUNT]) break;
    }
    const int* sg = hdr + HDR_SEG0 + s * SEG_WORDS;
    if (j < sg[SEG_K_COUNT]) {
      const int kid = hdr[HDR_KOFFS + sg[SEG_KOFF_BEGIN] + j];
      const int o = rows[p];
      const int idx = pcc_lookup(grid, in_keys, n_in, out_keys[o] - pcc_delta_of(kid, ks, step));
      const int64_t nb = ((int64_t)(unsigned)sg[SEG_NBR_LO]) | ((int64_t)sg[SEG_NBR_HI] << 32);
      nbr[nb + (int64_t)j * sg[SEG_POS_COUNT] + (p - sg[SEG_POS_BEGIN])] = idx;
      hit = idx >= 0;
    }
  }
  count_pairs(hit, d_pairs);
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static int max_class_k(int ks, int stride) {
  // offsets per axis congruent to a residue: max over residues
  int best = 0;
  for (int r = 0; r < stride; ++r) {
    int c = 0;
    for (int i = 0; i < ks; ++i) {
      const int off = (ks & 1) ? i - (ks - 1) / 2 : i;
      if (((off % stride) + stride) % stride == r) ++c;
    }
    if (c > best) best = c;
  }
  return best * best * best;
}

extern "C" int64_t pcc_map_nbr_elems(int64_t n_out, int32_t kernel_size, int32_t stride, int32_t transposed) {
  const int64_t K = (int64_t)kernel_size * kernel_size * kernel_size;
  if (!transposed) return n_out * K;
  return n_out * max_class_k(kernel_size, stride);
}

extern "C" size_t pcc_map_ws_bytes(int64_t n_out) {
  if (n_out <= 0) return 256;
  // class ids (2x) + class sort + one pair-count partial per build block (K <= 125 grid rows)
  return 2 * pcc_align_up((size_t)n_out * 8) + pcc_sort_ws_bytes(n_out) +
         pcc_align_up((size_t)pcc_cdiv(n_out, 256) * 128 * 4) + 1024;
}

extern "C" int pcc_kernel_map_build(const int64_t* in_keys, int64_t n_in, const int64_t* out_keys,
                                    int64_t n_out, int32_t kernel_size, int32_t step, int32_t stride,
                                    int32_t transposed, int32_t* hdr, int32_t* nbr, int32_t* rows,
                                    int64_t* d_pairs, const uint64_t* grid_bits, const int32_t* grid_rank,
                                    const int32_t* h_grid, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PccGrid grid;
  memset(&grid, 0, sizeof(grid));
  if (grid_bits) {
    PCC_REQUIRE(grid_rank && h_grid, "pcc_kernel_map_build: grid needs rank[] and its 8 host parameters");
    grid.bits = (const unsigned long long*)grid_bits;
    grid.rank = grid_rank;
    for (int i = 0; i < 3; ++i) { grid.lo[i] = h_grid[i]; grid.dims[i] = h_grid[3 + i]; }
    PCC_REQUIRE(h_grid[6] >= 1 && (h_grid[6] & (h_grid[6] - 1)) == 0, "pcc_kernel_map_build: grid pitch must be a power of two");
    int l = 0; while ((1 << l) < h_grid[6]) ++l;
    grid.ts_log2 = l;
    grid.nbatch = h_grid[7];
  }
  PCC_REQUIRE(hdr, "pcc_kernel_map_build: hdr is NULL");
  PCC_REQUIRE(kernel_size >= 1 && kernel_size <= 5, "pcc_kernel_map_build: kernel_size %d unsupported", kernel_size);
  PCC_REQUIRE(step >= 1 && (step & (step - 1)) == 0, "pcc_kernel_map_build: step %d is not a power of two", step);
  PCC_REQUIRE(n_in < (1ll << 31) && n_out < (1ll << 31), "pcc_kernel_map_build: too many rows");
  const int K = kernel_size * kernel_size * kernel_size;
  if (d_pairs) PCC_CHECK_HIP(hipMemsetAsync(d_pairs, 0, sizeof(int64_t), s));
  if (n_out > 0 && ws_bytes < pcc_map_ws_bytes(n_out)) {
    pcc_set_error("pcc_kernel_map_build: workspace too small");
    return PCC_EWS;
  }
  PCC_REQUIRE(n_out == 0 || ws, "pcc_kernel_map_build: ws is NULL");
  // pair-count partials live at the front of the workspace
  int* block_counts = d_pairs ? (int*)ws : nullptr;
  const size_t bc_bytes = pcc_align_up((size_t)pcc_cdiv(n_out > 0 ? n_out : 1, 256) * 128 * 4);
  SegPlan plan;
  memset(&plan, 0, sizeof(plan));
  plan.K = K;
  if (!transposed) {
    plan.nseg = 1;
    plan.listed = 0;
    plan.k_count[0] = K;
    plan.koff_begin[0] = 0;
    plan_order(plan, kernel_size);
    const bool hdr_in_map = n_out > 0 && grid.bits && (kernel_size == 3 || kernel_size == 5);
    if (!hdr_in_map) {
      k_write_hdr<<<1, 192, 0, s>>>(plan, nullptr, 0, n_out, hdr);
      PCC_LAUNCH_CHECK();
    }
    if (n_out == 0) return PCC_OK;
    PCC_REQUIRE(in_keys && out_keys && nbr, "pcc_kernel_map_build: NULL array");
    dim3 gdim((unsigned)pcc_cdiv(n_out, 256), (unsigned)K);
    const int* morton_rows = nullptr;
    if (rows) {   // conv map with positions in Morton (Z-curve) order of the output coordinates
      PCC_REQUIRE(ws && ws_bytes >= pcc_map_ws_bytes(n_out), "pcc_kernel_map_build: workspace too small");
      char* p = (char*)ws + bc_bytes;
      uint64_t* code = (uint64_t*)p;         p += pcc_align_up((size_t)n_out * 8);
      uint64_t* code_sorted = (uint64_t*)p;  p += pcc_align_up((size_t)n_out * 8) + 1024;
      k_morton<<<(unsigned)pcc_cdiv(n_out, 256), 256, 0, s>>>(out_keys, n_out, ilog2(step), code);
      PCC_LAUNCH_CHECK();
      PCC_TRY(pcc_sort_keys((const int64_t*)code, n_out, 0x7FFFFFFFFFFFFFFFull, (int64_t*)code_sorted, rows, p,
                            ws_bytes - (size_t)(p - (char*)ws), s));
      morton_rows = rows;
    }
    if (grid.bits && (kernel_size == 3 || kernel_size == 5)) {
      gdim.y = (unsigned)(kernel_size * kernel_size);
      if (kernel_size == 3) k_map_conv_z<3><<<gdim, 256, 0, s>>>(grid, out_keys, n_out, step, morton_rows, nbr, block_counts, plan, hdr);
      else k_map_conv_z<5><<<gdim, 256, 0, s>>>(grid, out_keys, n_out, step, morton_rows, nbr, block_counts, plan, hdr);
    } else {
      k_map_conv<<<gdim, 256, 0, s>>>(grid, in_keys, (int)n_in, out_keys, n_out, kernel_size, step, morton_rows, nbr,
                                      block_counts);
    }
    PCC_LAUNCH_CHECK();
    if (d_pairs) {
      k_sum_counts<<<1, 1024, 0, s>>>(block_counts, (int64_t)gdim.x * gdim.y, d_pairs);
      PCC_LAUNCH_CHECK();
    }
    return PCC_OK;
  }
  // ---- transposed ------------------------------------------------------------------------------
  PCC_REQUIRE(stride == 1 || stride == 2, "pcc_kernel_map_build: transposed stride %d unsupported", stride);
  const int nclass = stride * stride * stride;
  plan.nseg = nclass;
  plan.listed = 1;
  int fill = 0;
  for (int c = 0; c < nclass; ++c) {
    const int rx = c % stride, ry = (c / stride) % stride, rz = c / (stride * stride);
    plan.koff_begin[c] = fill;
    for (int kid = 0; kid < K; ++kid) {
      int dx, dy, dz;
      pcc_offset_of(kid, kernel_size, dx, dy, dz);
      auto md = [&](int v) { return ((v % stride) + stride) % stride; };
      if (md(dx) == rx && md(dy) == ry && md(dz) == rz) plan.koffs[fill++] = (unsigned char)kid;
    }
    plan.k_count[c] = fill - plan.koff_begin[c];
  }
  plan_order(plan, kernel_size);
  if (n_out == 0) {
    k_write_hdr<<<1, 192, 0, s>>>(plan, nullptr, 0, 0, hdr);   // all segments empty
    PCC_LAUNCH_CHECK();
    return PCC_OK;
  }
  PCC_REQUIRE(in_keys && out_keys && nbr && rows, "pcc_kernel_map_build: NULL array (transposed needs rows)");
  char* p = (char*)ws + bc_bytes;
  uint64_t* cls = (uint64_t*)p;         p += pcc_align_up((size_t)n_out * 8);
  uint64_t* cls_sorted = (uint64_t*)p;  p += pcc_align_up((size_t)n_out * 8);
  int* class_begin = (int*)p;           p += 1024;
  void* sort_ws = p;
  dim3 g1((unsigned)pcc_cdiv(n_out, 256));
  k_classify<<<g1, 256, 0, s>>>(out_keys, n_out, ilog2(step), stride, cls);
  PCC_LAUNCH_CHECK();
  // stable counting sort by class == one radix pass; payload = output row of each position
  PCC_TRY(pcc_sort_keys((const int64_t*)cls, n_out, 0xFFull, (int64_t*)cls_sorted, rows, sort_ws,
                        ws_bytes - (size_t)(p - (char*)ws), s));
  k_class_begin<<<1, 64, 0, s>>>(cls_sorted, n_out, nclass, class_begin);
  PCC_LAUNCH_CHECK();
  k_write_hdr<<<1, 192, 0, s>>>(plan, class_begin, 1, n_out, hdr);
  PCC_LAUNCH_CHECK();
  dim3 gdim((unsigned)pcc_cdiv(n_out, 256), (unsigned)max_class_k(kernel_size, stride));
  k_map_transposed<<<gdim, 256, 0, s>>>(grid, in_keys, (int)n_in, out_keys, n_out, kernel_size, step, hdr, rows, nbr,
                                        block_counts);
  PCC_LAUNCH_CHECK();
  if (d_pairs) {
    k_sum_counts<<<1, 1024, 0, s>>>(block_counts, (int64_t)gdim.x * gdim.y, d_pairs);
    PCC_LAUNCH_CHECK();
  }
  return PCC_OK;
}

// ---- grid index build ---------------------------------------------------------------------------------
static int grid_rank(const unsigned long long* b, int64_t words, int32_t* rank, const int32_t* h, int tsl, int64_t* out_keys,
                     int64_t* d_count, void* ws, size_t ws_bytes, hipStream_t s);
__device__ inline long long grid_cell(const int64_t key, const int lo0, const int lo1, const int lo2, const int d0,
                                      const int d1, const int d2, const int tsl) {
  const int b = (int)(key >> 48);
  const int cx = ((int)((key >> 32) & 0xFFFF) - (int)PCC_BIAS - lo0) >> tsl;
  const int cy = ((int)((key >> 16) & 0xFFFF) - (int)PCC_BIAS - lo1) >> tsl;
  const int cz = ((int)(key & 0xFFFF) - (int)PCC_BIAS - lo2) >> tsl;
  return (((long long)b * d0 + cx) * d1 + cy) * d2 + cz;
}

// keys are sorted => cells ascending: the first row of each word ORs the bits of its (<= 64) followers; no atomics
__global__ void k_grid_bits(const int64_t* __restrict__ keys, int64_t n, int lo0, int lo1, int lo2, int d0, int d1,
                            int d2, int tsl, unsigned long long* __restrict__ bits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long c = grid_cell(keys[i], lo0, lo1, lo2, d0, d1, d2, tsl);
  const long long w = c >> 6;
  if (i > 0 && (grid_cell(keys[i - 1], lo0, lo1, lo2, d0, d1, d2, tsl) >> 6) == w) return;
  unsigned long long acc = 1ull << (c & 63);
  for (int64_t t = i + 1; t < n && t < i + 64; ++t) {
    const long long c2 = grid_cell(keys[t], lo0, lo1, lo2, d0, d1, d2, tsl);
    if ((c2 >> 6) != w) break;
    acc |= 1ull << (c2 & 63);
  }
  bits[w] = acc;
}

__global__ void k_grid_popc(const unsigned long long* __restrict__ bits, int64_t words, int* __restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < words) cnt[i] = __popcll(bits[i]);
}

extern "C" int64_t pcc_grid_words(const int32_t* h_grid) {
  const long long cells = (long long)h_grid[7] * h_grid[3] * h_grid[4] * h_grid[5];
  // an EVEN number of 64-bit words: hipMemsetAsync clears a size that is not a multiple of 16 bytes with two kernels (bulk + an
  // 8-byte tail, each a launch: tools/probes/memset_probe.hip), and a step clears ~20 bitmaps
  return ((cells + 63) / 64 + 1) / 2 * 2;
}

extern "C" size_t pcc_grid_ws_bytes(int64_t words) { return pcc_scan_ws_bytes(words) + 256; }

extern "C" int pcc_grid_build(const int64_t* keys, int64_t n, const int32_t* h_grid, uint64_t* bits, int32_t* rank,
                              void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(h_grid && bits && rank, "pcc_grid_build: NULL array");
  const int64_t words = pcc_grid_words(h_grid);
  PCC_REQUIRE(words >= 1 && words < (1ll << 31), "pcc_grid_build: lattice too large (%lld words)", (long long)words);
  PCC_REQUIRE(h_grid[6] >= 1 && (h_grid[6] & (h_grid[6] - 1)) == 0, "pcc_grid_build: pitch must be a power of two");
  if (ws_bytes < pcc_grid_ws_bytes(words)) {
    pcc_set_error("pcc_grid_build: workspace too small");
    return PCC_EWS;
  }
  PCC_CHECK_HIP(hipMemsetAsync(bits, 0, (size_t)words * 8, s));
  if (n > 0) {
    PCC_REQUIRE(keys, "pcc_grid_build: keys is NULL");
    k_grid_bits<<<(unsigned)pcc_cdiv(n, 256), 256, 0, s>>>(keys, n, h_grid[0], h_grid[1], h_grid[2], h_grid[3], h_grid[4],
                                                          h_grid[5], ilog2(h_grid[6]), (unsigned long long*)bits);
    PCC_LAUNCH_CHECK();
  }
  return grid_rank((const unsigned long long*)bits, words, rank, h_grid, 0, nullptr, nullptr, ws, ws_bytes, s);
}

// ---- strided coordinate set straight from the occupancy bitmap --------------------------------------------
// unique(floor(c/m)*m) needs no sort: mark the coarse cell of every fine row (atomicOr; fine order is not coarse cell
// order), rank the words, and read the set back out of the bitmap -- bitmap order IS canonical key order.  The coarse
// set's grid index (bits + rank) falls out for free.  A handful of launches where the sort took ~35.
// d_n (nullable): the row count lives on the device (a set whose size the host has not read yet); n is then the capacity
// the grid was sized for, and threads past *d_n leave.
__global__ void k_grid_bits_any(const int64_t* __restrict__ keys, int64_t n, const int64_t* __restrict__ d_n, int lo0, int lo1,
                                int lo2, int d0, int d1, int d2, int tsl, unsigned long long* __restrict__ bits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (d_n ? *d_n : n)) return;
  const long long c = grid_cell(keys[i], lo0, lo1, lo2, d0, d1, d2, tsl);
  // neighbours in fine order often share the coarse cell (z pairs): only the first of a run issues the atomic
  if (i > 0 && grid_cell(keys[i - 1], lo0, lo1, lo2, d0, d1, d2, tsl) == c) return;
  // many fine rows per coarse cell (user-ordered rows, large pitch ratios): a cell already marked needs no atomic.  The
  // load goes to L2 (agent scope), so it sees the marks of other CUs; a stale 0 only costs a redundant atomicOr.
  const unsigned long long m = 1ull << (c & 63);
  if (__hip_atomic_load(&bits[c >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & m) return;
  atomicOr(&bits[c >> 6], m);
}

__global__ void k_grid_enumerate(const unsigned long long* __restrict__ bits, const int* __restrict__ rank,
                                 int64_t words, int lo0, int lo1, int lo2, int d0, int d1, int d2, int tsl,
                                 int64_t* __restrict__ keys, int64_t* __restrict__ count) {
  const int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (wi >= words) return;
  unsigned long long w = bits[wi];
  int r = rank[wi];
  if (wi == words - 1) *count = (int64_t)r + __popcll(w);
  if (!w) return;
  // cell of the word's first bit, decomposed once (three 64-bit divisions per WORD, not per set bit); a bit's z may run past
  // the end of the z axis and carry into y / x / batch
  long long cell = wi * 64;
  const int z0 = (int)(cell % d2); cell /= d2;
  const int y0 = (int)(cell % d1); cell /= d1;
  const int x0 = (int)(cell % d0);
  const int b0 = (int)(cell / d0);
  while (w) {
    const int bit = __ffsll((long long)w) - 1;
    w &= w - 1;
    int cz = z0 + bit, cy = y0, cx = x0, b = b0;
    while (cz >= d2) {
      cz -= d2;
      if (++cy == d1) { cy = 0; if (++cx == d0) { cx = 0; ++b; } }
    }
    const int64_t x = (int64_t)(lo0 + (cx << tsl)) + PCC_BIAS, y = (int64_t)(lo1 + (cy << tsl)) + PCC_BIAS,
                  z = (int64_t)(lo2 + (cz << tsl)) + PCC_BIAS;
    keys[r++] = ((int64_t)b << 48) | (x << 32) | (y << 16) | z;
  }
}

// rank (+ read-out) of a bitmap in two launches (round 3; popcount kernel + scan (2 launches) + read-out kernel before, each
// a pass over the words of the whole bounding lattice -- 16.8 M words for the last level's candidates): the reduce pass
// popcounts the words itself, the apply pass derives its workgroup's offset from the workgroup totals, writes rank[] and reads
// the set bits back out of the words it already holds.  2048 words per workgroup (256 threads x 8 consecutive words).
static constexpr int GR_B = 2048;
static constexpr int64_t GR_DIRECT_NB = 16384;          // workgroup totals the apply pass sums up itself (as the scan does)
static constexpr int64_t GR_SELF_NB = 8;                // lattices of <= 8 x 2048 words (1 M cells): rank + read-out in ONE launch

__global__ void __launch_bounds__(256) k_grid_rank_reduce(const unsigned long long* __restrict__ bits, int64_t words,
                                                          int* __restrict__ sums) {
  __shared__ int wsum[4];
  const int64_t base = (int64_t)blockIdx.x * GR_B + (int64_t)threadIdx.x * 8;
  int c = 0;
  if (base + 8 <= words) {
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(bits + base);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const ulonglong2 v = p[q]; c += __popcll(v.x) + __popcll(v.y); }
  } else {
    for (int q = 0; q < 8; ++q) if (base + q < words) c += __popcll(bits[base + q]);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// SELF: no reduce pass ran -- the workgroup popcounts the words of the (few) workgroups before it itself.  Small lattices only
// (<= GR_SELF_NB workgroups): there the three launches reduce / apply / read-out are pure launch latency, ~5 us each, and a step
// builds about ten such grids (the hyper-prior's sets, the coarse end of the stride chain); ONE launch does all three.
template <bool ENUM, bool SELF = false>
__global__ void __launch_bounds__(256) k_grid_rank_apply(const unsigned long long* __restrict__ bits, int64_t words,
                                                         const int* __restrict__ sums, int* __restrict__ rank, int lo0, int lo1,
                                                         int lo2, int d0, int d1, int d2, int tsl, int64_t* __restrict__ keys,
                                                         int64_t* __restrict__ count) {
  __shared__ int wsum[4];
  __shared__ int s_off;
  __shared__ int rk_s[ENUM ? GR_B : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // offset of this workgroup: totals of the workgroups before it
  int part = 0;
  if (SELF) { for (int64_t i = threadIdx.x; i < (int64_t)blockIdx.x * GR_B; i += 256) part += __popcll(bits[i]); }
  else
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) part += sums[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
  if (lane == 0) wsum[wv] = part;
  __syncthreads();
  if (threadIdx.x == 0) s_off = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  const int off = s_off;
  const int64_t base = (int64_t)blockIdx.x * GR_B + (int64_t)threadIdx.x * 8;
  unsigned long long w[8];
  if (base + 8 <= words) {
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(bits + base);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const ulonglong2 v = p[q]; w[2 * q] = v.x; w[2 * q + 1] = v.y; }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = base + q < words ? bits[base + q] : 0ull;
  }
  int c = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) c += __popcll(w[q]);
  int inc = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
  __syncthreads();
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  int run = off + inc - c;
  for (int q = 0; q < wv; ++q) run += wsum[q];
  int r8[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { r8[q] = run; run += __popcll(w[q]); }
  if (base + 8 <= words) {
    reinterpret_cast<int4*>(rank + base)[0] = make_int4(r8[0], r8[1], r8[2], r8[3]);
    reinterpret_cast<int4*>(rank + base)[1] = make_int4(r8[4], r8[5], r8[6], r8[7]);
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) if (base + q < words) rank[base + q] = r8[q];
  }
  if (!ENUM) return;
  if (base <= words - 1 && words - 1 < base + 8) *count = (int64_t)run;          // the thread holding the last word: total
  // read-out, one word per thread and trip (adjacent threads = adjacent words, so their key runs are adjacent in memory)
#pragma unroll
  for (int q = 0; q < 8; ++q) rk_s[threadIdx.x * 8 + q] = r8[q];
  __syncthreads();
  const int total_here = (wsum[0] + wsum[1] + wsum[2] + wsum[3]);
  if (total_here == 0) return;                                                    // empty stretch of the lattice
  for (int t = 0; t < 8; ++t) {
    const int li = t * 256 + (int)threadIdx.x;
    const int64_t wi = (int64_t)blockIdx.x * GR_B + li;
    if (wi >= words) break;
    unsigned long long x = bits[wi];                                              // (L1 / L2 hit: this workgroup just read it)
    if (!x) continue;
    int r = rk_s[li];
    long long cell = wi * 64;
    const int z0 = (int)(cell % d2); cell /= d2;
    const int y0 = (int)(cell % d1); cell /= d1;
    const int x0 = (int)(cell % d0);
    const int b0 = (int)(cell / d0);
    while (x) {
      const int bit = __ffsll((long long)x) - 1;
      x &= x - 1;
      int cz = z0 + bit, cy = y0, cx = x0, b = b0;
      while (cz >= d2) {
        cz -= d2;
        if (++cy == d1) { cy = 0; if (++cx == d0) { cx = 0; ++b; } }
      }
      const int64_t X = (int64_t)(lo0 + (cx << tsl)) + PCC_BIAS, Y = (int64_t)(lo1 + (cy << tsl)) + PCC_BIAS,
                    Z = (int64_t)(lo2 + (cz << tsl)) + PCC_BIAS;
      keys[r++] = ((int64_t)b << 48) | (X << 32) | (Y << 16) | Z;
    }
  }
}

static int g_grid_rank_fused = getenv("PCC_GRID_RANK_FUSED") ? atoi(getenv("PCC_GRID_RANK_FUSED")) : 1;   // 2: read-out inside the apply pass

// rank[] of a marked bitmap and, with out_keys, the canonical keys of its set bits + their count.  ws: pcc_grid_ws_bytes(words).
static int grid_rank(const unsigned long long* b, int64_t words, int32_t* rank, const int32_t* h, int tsl, int64_t* out_keys,
                     int64_t* d_count, void* ws, size_t ws_bytes, hipStream_t s) {
  const int64_t nb = pcc_cdiv(words, GR_B);
  if (g_grid_rank_fused && nb <= GR_SELF_NB && (((uintptr_t)b | (uintptr_t)rank) & 15) == 0) {      // small lattice: one launch
    if (out_keys) k_grid_rank_apply<true, true><<<(unsigned)nb, 256, 0, s>>>(b, words, nullptr, rank, h[0], h[1], h[2], h[3], h[4], h[5], tsl, out_keys, d_count);
    else k_grid_rank_apply<false, true><<<(unsigned)nb, 256, 0, s>>>(b, words, nullptr, rank, 0, 0, 0, 1, 1, 1, 0, nullptr, nullptr);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
  }
  if (g_grid_rank_fused && nb <= GR_DIRECT_NB && ws_bytes >= (size_t)nb * 4 + 256 && (((uintptr_t)b | (uintptr_t)rank) & 15) == 0) {
    int* sums = (int*)ws;
    k_grid_rank_reduce<<<(unsigned)nb, 256, 0, s>>>(b, words, sums);
    // (the read-out inside the apply pass -- k_grid_rank_apply<true>, 8 words per thread -- measured 2x slower than the
    //  word-per-thread read-out kernel: 0.61 against 0.27 ms per step; the rank passes alone save the popcount kernel and
    //  its 200 MB round trip on the large lattices)
    if (out_keys && g_grid_rank_fused > 1) {
      k_grid_rank_apply<true><<<(unsigned)nb, 256, 0, s>>>(b, words, sums, rank, h[0], h[1], h[2], h[3], h[4], h[5], tsl, out_keys, d_count);
      PCC_LAUNCH_CHECK();
      return PCC_OK;
    }
    k_grid_rank_apply<false><<<(unsigned)nb, 256, 0, s>>>(b, words, sums, rank, 0, 0, 0, 1, 1, 1, 0, nullptr, nullptr);
    PCC_LAUNCH_CHECK();
    if (out_keys) {
      k_grid_enumerate<<<(unsigned)pcc_cdiv(words, 256), 256, 0, s>>>(b, rank, words, h[0], h[1], h[2], h[3], h[4], h[5], tsl, out_keys, d_count);
      PCC_LAUNCH_CHECK();
    }
    return PCC_OK;
  }
  k_grid_popc<<<(unsigned)pcc_cdiv(words, 256), 256, 0, s>>>(b, words, rank);
  PCC_LAUNCH_CHECK();
  PCC_TRY(pcc_scan_exclusive_i32(rank, rank, words, ws, ws_bytes, s));
  if (out_keys) {
    k_grid_enumerate<<<(unsigned)pcc_cdiv(words, 256), 256, 0, s>>>(b, rank, words, h[0], h[1], h[2], h[3], h[4], h[5], tsl, out_keys, d_count);
    PCC_LAUNCH_CHECK();
  }
  return PCC_OK;
}

// keys: the FINE set (canonical); h_grid: lattice of the COARSE set (lo multiples of the coarse pitch h_grid[6]).
// Outputs: bits/rank = grid index of the coarse set, out_keys (capacity n) = its canonical keys, *d_count = its size.
extern "C" int pcc_coords_stride_grid(const int64_t* keys, int64_t n, const int64_t* d_n, const int32_t* h_grid, uint64_t* bits,
                                      int32_t* rank, int64_t* out_keys, int64_t* d_count, void* ws, size_t ws_bytes,
                                      void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(h_grid && bits && rank && out_keys && d_count, "pcc_coords_stride_grid: NULL array");
  const int64_t words = pcc_grid_words(h_grid);
  PCC_REQUIRE(words >= 1 && words < (1ll << 31), "pcc_coords_stride_grid: lattice too large (%lld words)", (long long)words);
  const int P = h_grid[6];
  PCC_REQUIRE(P >= 1 && (P & (P - 1)) == 0, "pcc_coords_stride_grid: pitch must be a power of two");
  PCC_REQUIRE(h_grid[0] % P == 0 && h_grid[1] % P == 0 && h_grid[2] % P == 0,
              "pcc_coords_stride_grid: lattice origin must be a multiple of the pitch %d", P);
  if (ws_bytes < pcc_grid_ws_bytes(words)) {
    pcc_set_error("pcc_coords_stride_grid: workspace too small");
    return PCC_EWS;
  }
  PCC_CHECK_HIP(hipMemsetAsync(bits, 0, (size_t)words * 8, s));
  if (n > 0) {
    PCC_REQUIRE(keys, "pcc_coords_stride_grid: keys is NULL");
    k_grid_bits_any<<<(unsigned)pcc_cdiv(n, 256), 256, 0, s>>>(keys, n, d_n, h_grid[0], h_grid[1], h_grid[2], h_grid[3],
                                                              h_grid[4], h_grid[5], ilog2(P), (unsigned long long*)bits);
    PCC_LAUNCH_CHECK();
  }
  return grid_rank((const unsigned long long*)bits, words, rank, h_grid, ilog2(P), out_keys, d_count, ws, ws_bytes, s);
}

// ---- canonical order of user-ordered rows through the bitmap (ME.SparseTensor construction, a1) ------------------
__global__ void k_grid_bits_each(const int64_t* __restrict__ keys, int64_t n, int lo0, int lo1, int lo2, int d0, int d1,
                                 int d2, int tsl, unsigned long long* __restrict__ bits, int64_t* __restrict__ off_lattice) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t key = keys[i];
  const int x = (int)((key >> 32) & 0xFFFF) - (int)PCC_BIAS - lo0, y = (int)((key >> 16) & 0xFFFF) - (int)PCC_BIAS - lo1,
            z = (int)(key & 0xFFFF) - (int)PCC_BIAS - lo2;
  if ((x | y | z) & ((1 << tsl) - 1)) { *off_lattice = 1; return; }      // not a multiple of the pitch: caller falls back
  const long long c = grid_cell(key, lo0, lo1, lo2, d0, d1, d2, tsl);
  atomicOr(&bits[c >> 6], 1ull << (c & 63));
}

// canonical position of every user row; the smallest user row of a cell wins (first duplicate kept)
__global__ void k_grid_first_user(const int64_t* __restrict__ keys, int64_t n, int lo0, int lo1, int lo2, int d0, int d1,
                                  int d2, int tsl, const unsigned long long* __restrict__ bits,
                                  const int* __restrict__ rank, int* __restrict__ first_user) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long c = grid_cell(keys[i], lo0, lo1, lo2, d0, d1, d2, tsl);
  const unsigned long long w = bits[c >> 6];
  const int r = rank[c >> 6] + __popcll(w & ((1ull << (c & 63)) - 1ull));
  atomicMin(&first_user[r], (int)i);
}

// keys in USER order (any order, duplicates allowed).  Outputs: the set's grid index (bits, rank), its canonical keys
// (capacity n), first_user[canonical position] = smallest user row holding that coordinate, d_count[0] = unique rows,
// d_count[1] != 0: some key is not on the lattice (not a multiple of the pitch) -- results invalid, use the sort path.
extern "C" int pcc_keys_canonicalize_grid(const int64_t* keys, int64_t n, const int32_t* h_grid, uint64_t* bits,
                                          int32_t* rank, int64_t* out_keys, int32_t* first_user, int64_t* d_count,
                                          void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(keys && h_grid && bits && rank && out_keys && first_user && d_count && n > 0 && n < (1ll << 31),
              "pcc_keys_canonicalize_grid: bad arguments");
  const int64_t words = pcc_grid_words(h_grid);
  PCC_REQUIRE(words >= 1 && words < (1ll << 31), "pcc_keys_canonicalize_grid: lattice too large (%lld words)", (long long)words);
  const int P = h_grid[6];
  PCC_REQUIRE(P >= 1 && (P & (P - 1)) == 0, "pcc_keys_canonicalize_grid: pitch must be a power of two");
  if (ws_bytes < pcc_grid_ws_bytes(words)) { pcc_set_error("pcc_keys_canonicalize_grid: workspace too small"); return PCC_EWS; }
  const int tsl = ilog2(P);
  unsigned long long* b = (unsigned long long*)bits;
  PCC_CHECK_HIP(hipMemsetAsync(bits, 0, (size_t)words * 8, s));
  PCC_CHECK_HIP(hipMemsetAsync(first_user, 0x7F, (size_t)n * 4, s));
  PCC_CHECK_HIP(hipMemsetAsync(d_count, 0, 2 * sizeof(int64_t), s));
  const unsigned g = (unsigned)pcc_cdiv(n, 256);
  k_grid_bits_each<<<g, 256, 0, s>>>(keys, n, h_grid[0], h_grid[1], h_grid[2], h_grid[3], h_grid[4], h_grid[5], tsl, b,
                                     d_count + 1);
  PCC_LAUNCH_CHECK();
  PCC_TRY(grid_rank(b, words, rank, h_grid, tsl, out_keys, d_count, ws, ws_bytes, s));
  k_grid_first_user<<<g, 256, 0, s>>>(keys, n, h_grid[0], h_grid[1], h_grid[2], h_grid[3], h_grid[4], h_grid[5], tsl, b,
                                      rank, first_user);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ---- generative expansion through the bitmaps ----------------------------------------------------------------
// out = union_k (in + off_k * ts_out), plus the transposed map as CSR pair lists, without sorting the n*K candidates:
//   mark   : thread (input row, kernel column kx,ky) ORs its ks z-consecutive cells into the output bitmap (<= 2 atomics)
//   rank   : popcount + scan; enumerate the set bits -> canonical output keys (and the output set's grid index)
//   csr    : thread per output row probes the INPUT grid at c - off_k for the offsets whose parity fits, x then y then z
//            descending, i.e. in ascending input row: exactly the pair-id order a stable sort by cell would give.
//            One pass counts, a scan gives first[], a second pass writes pair ids i*K + k.
template <int KS>
__global__ void __launch_bounds__(256) k_expand_mark(const int64_t* __restrict__ keys, int64_t n, int ts_out, int lo0,
                                                     int lo1, int lo2, int d0, int d1, int d2, int tsl,
                                                     unsigned long long* __restrict__ bits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int H = (KS & 1) ? (KS - 1) / 2 : 0;
  const int kxy = blockIdx.y;
  const int64_t key = keys[i];
  const int b = (int)(key >> 48);
  const int cx = ((int)((key >> 32) & 0xFFFF) - (int)PCC_BIAS - lo0 + (kxy % KS - H) * ts_out) >> tsl;
  const int cy = ((int)((key >> 16) & 0xFFFF) - (int)PCC_BIAS - lo1 + (kxy / KS - H) * ts_out) >> tsl;
  const int cz = ((int)(key & 0xFFFF) - (int)PCC_BIAS - lo2 - H * ts_out) >> tsl;
  const long long cell = (((long long)b * d0 + cx) * d1 + cy) * d2 + cz;     // KS consecutive cells along z
  const int bit = (int)(cell & 63);
  const unsigned long long m = (1ull << KS) - 1ull;
  atomicOr(&bits[cell >> 6], m << bit);
  if (bit + KS > 64) atomicOr(&bits[(cell >> 6) + 1], m >> (64 - bit));
}

struct ExpandCsrArgs {
  const int64_t* out_keys; long long n_out;
  PccGrid in;                 // grid index of the input set
  int ts_out; int K;
  int* first;                 // FILL=false: first[o] = pairs of row o;  FILL=true: exclusive prefix (read)
  int* pair_ids;
  int zk;                     // kernel-offset index of a pair: 0 = x fastest (ix + KS*iy + KS*KS*iz, the native order), 1 = z fastest
  long long* d_total;         // nullable: receives first[n_out], the number of pairs (fill pass)
};

// The compatible source cells of an output row: per axis the offsets whose source lies on the input lattice (offset index
// descending = source cell ascending).  They are CONSECUTIVE lattice cells per axis, so the z cells of a kernel column are
// one bit field of <= 4 bits in the occupancy bitmap (at most two words): a column costs one field extraction instead of
// a loop over its cells -- the per-cell loops were instruction-bound (64 divergent iterations per row, 0.9 + 2.1 ms per step
// for the 7-wide lists of the composite convolutions in round 2).
// (Scalars, not per-cell arrays: the cells of an axis are consecutive and their offset indices step down by the pitch ratio,
// so cell j is c0 + j with index i0 - step * j.  Arrays indexed by a loop counter went to scratch memory -- 188 bytes per
// thread, 2.7 GB of scratch traffic per pass over the last level's 14.5 M rows.)
// (Scalar members, round 3: as arrays c0[3] / i0[3] / cnt[3] the z entries still went through scratch memory -- a store and a
//  dependent reload at the top of every thread of the count and fill passes.)
template <int KS>
struct CsrCols {
  int c0x, c0y, c0z, i0x, i0y, i0z, nx, ny, nz;
  int step, b;
  __device__ __forceinline__ int cell(int ax, int j) const { return (ax == 0 ? c0x : ax == 1 ? c0y : c0z) + j; }
  __device__ __forceinline__ int idx(int ax, int j) const { return (ax == 0 ? i0x : ax == 1 ? i0y : i0z) - step * j; }
  __device__ __forceinline__ int cnt(int ax) const { return ax == 0 ? nx : ax == 1 ? ny : nz; }
  __device__ __forceinline__ void set(int ax, int c0, int i0, int n) {
    if (ax == 0) { c0x = c0; i0x = i0; nx = n; } else if (ax == 1) { c0y = c0; i0y = i0; ny = n; } else { c0z = c0; i0z = i0; nz = n; }
  }
};

template <int KS>
__device__ __forceinline__ void csr_columns(const ExpandCsrArgs& a, long long o, CsrCols<KS>& c) {
  constexpr int H = (KS & 1) ? (KS - 1) / 2 : 0;
  const int64_t key = a.out_keys[o];
  c.b = (int)(key >> 48);
  const int p[3] = {(int)((key >> 32) & 0xFFFF) - (int)PCC_BIAS - a.in.lo[0],
                    (int)((key >> 16) & 0xFFFF) - (int)PCC_BIAS - a.in.lo[1],
                    (int)(key & 0xFFFF) - (int)PCC_BIAS - a.in.lo[2]};
  const int tm = (1 << a.in.ts_log2) - 1;
  c.step = (1 << a.in.ts_log2) / a.ts_out;
  if ((KS & 1) && a.ts_out * 2 == (1 << a.in.ts_log2)) {
    // up-sampling by 2 with an odd kernel (the generative convolutions of the codec): in units of the output pitch the
    // source cell cc holds offset index ia = P + H - 2 cc, so the compatible cells are the run
    // [ceil((P - H) / 2), floor((P + H) / 2)] cut to the lattice -- closed form instead of a scan over the KS offsets
    const int tl = a.in.ts_log2 - 1;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const int P = p[ax] >> tl;                         // (p is a multiple of the output pitch)
      int c0 = (P - H + 1) >> 1, c1 = (P + H) >> 1;
      c0 = c0 < 0 ? 0 : c0;
      c1 = c1 >= a.in.dims[ax] ? a.in.dims[ax] - 1 : c1;
      const int m = c1 - c0 + 1;
      c.set(ax, c0, P + H - 2 * c0, (m > 0 && !(p[ax] & (a.ts_out - 1))) ? m : 0);      // (rows off the output lattice have no source)
    }
    return;
  }
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    int m = 0, c0 = 0, i0 = 0;
#pragma unroll
    for (int ia = KS - 1; ia >= 0; --ia) {
      const int rel = p[ax] - (ia - H) * a.ts_out;
      const int cc = rel >> a.in.ts_log2;
      if (rel >= 0 && !(rel & tm) && cc < a.in.dims[ax]) {
        if (m == 0) { c0 = cc; i0 = ia; }
        ++m;
      }
    }
    c.set(ax, c0, i0, m);
  }
}

// occupancy bits of the nz consecutive cells starting at `cell` (nz <= 8); *word = index of the first word
__device__ __forceinline__ unsigned csr_field(const unsigned long long* __restrict__ bits, long long cell, int nz, long long* word,
                                              unsigned long long* w0_out) {
  const long long wi = cell >> 6;
  const int sh = (int)(cell & 63);
  const unsigned long long w0 = bits[wi];
  unsigned long long f = w0 >> sh;
  if (sh + nz > 64) f |= bits[wi + 1] << (64 - sh);
  *word = wi;
  *w0_out = w0;
  return (unsigned)f & ((1u << nz) - 1u);
}

// count pass: pairs of every output row.  M = compatible offsets per axis the probe covers: (KS + 1) / 2 when the input pitch
// is at least twice the output pitch (every generative convolution of the codec), KS for a stride-1 generative transpose
// (input pitch == output pitch: every offset is compatible).
template <int KS, int M>
__global__ void __launch_bounds__(256) k_expand_csr_count(ExpandCsrArgs a) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= a.n_out) return;
  CsrCols<KS> c;
  csr_columns<KS>(a, o, c);
  int total = 0;
  if (c.b < a.in.nbatch && c.cnt(2) > 0) {
    // All (x, y) columns at once: a column's z field (<= 4 bits) is cut out of the 64-bit window that starts at the 32-bit
    // word holding its first cell -- it never straddles, so a column is ONE unconditional load (absent columns re-read cell 0
    // and are masked) and the <= 16 loads of a row are in flight together.  (The loop form waited for each column's word.)
    const unsigned* const bits32 = reinterpret_cast<const unsigned*>(a.in.bits);
    const long long cells = (long long)a.in.nbatch * a.in.dims[0] * a.in.dims[1] * a.in.dims[2];
    const long long last_dw = 2 * ((cells + 63) >> 6) - 2;       // last 32-bit word a 64-bit window may start at
    const unsigned fmask = (1u << c.cnt(2)) - 1u;
    // (cell of column (jx, jy) = cell of column (0, 0) + jx slabs + jy columns: one multiply chain per row, not per column)
    const long long cell00 = (((long long)c.b * a.in.dims[0] + c.cell(0, 0)) * a.in.dims[1] + c.cell(1, 0)) * a.in.dims[2] + c.cell(2, 0);
    const long long slab = (long long)a.in.dims[1] * a.in.dims[2];
    // two x slabs (2 M windows) per trip: 8 loads in flight at 7-wide lists, and half the registers of all 16 at once
    // (129 VGPRs = 3 waves per SIMD before; the pass is latency-bound, occupancy is what hides it)
    constexpr int G = M >= 4 ? 2 : M;
#pragma unroll 1
    for (int jx0 = 0; jx0 < M; jx0 += G) {
      unsigned long long w[G * M];
      int sh[G * M];
#pragma unroll
      for (int gx = 0; gx < G; ++gx)
#pragma unroll
        for (int jy = 0; jy < M; ++jy) {
          const int jx = jx0 + gx;
          const bool ok = jx < c.cnt(0) && jy < c.cnt(1);
          const long long cell = ok ? cell00 + jx * slab + (long long)jy * a.in.dims[2] : 0ll;
          const long long dw = cell >> 5, dw2 = dw < last_dw ? dw : last_dw;
          sh[gx * M + jy] = ok ? (int)(cell & 31) + 32 * (int)(dw - dw2) : 64;
          const unsigned lo = bits32[dw2], hi = bits32[dw2 + 1];
          w[gx * M + jy] = (unsigned long long)lo | ((unsigned long long)hi << 32);
        }
#pragma unroll
      for (int i = 0; i < G * M; ++i) total += sh[i] < 64 ? __popc((unsigned)(w[i] >> (sh[i] & 63)) & fmask) : 0;
    }
  }
  a.first[o] = total;
  if (o == a.n_out - 1) a.first[a.n_out] = 0;         // the scan runs over n_out + 1 entries (its last output = the pair total)
}

// on_hit(i, kidx) for every existing source of row o, ascending source cell (= ascending input row)
template <int KS, typename F>
__device__ __forceinline__ void csr_probe(const ExpandCsrArgs& a, long long o, F&& on_hit) {
  CsrCols<KS> c;
  csr_columns<KS>(a, o, c);
  if (c.b >= a.in.nbatch || c.cnt(2) == 0) return;
  // The y columns of one x slab together (round 3): per column the aligned 64-bit word of its first cell, the 32 bits after it
  // (a field of <= 4 bits never reaches past them) and the word's rank -- three unconditional loads, absent columns re-read
  // cell 0 and are masked -- so a slab's <= 12-15 loads are in flight at once.  (The column-by-column loop waited for each
  // column's word, and for its rank behind a branch: 16 exposed L2 latencies per row of the 7-wide lists.)
  constexpr int MY = KS == 7 ? 4 : KS;                 // compatible offsets per axis: <= (KS + 1) / 2 at a pitch ratio of 2, KS at 1
  const unsigned* const bits32 = reinterpret_cast<const unsigned*>(a.in.bits);
  const long long cells = (long long)a.in.nbatch * a.in.dims[0] * a.in.dims[1] * a.in.dims[2];
  const long long last_w = ((cells + 63) >> 6) - 1;
  const unsigned fmask = (1u << c.cnt(2)) - 1u;
  const int kzs = a.zk ? 1 : KS * KS;
  const long long cell00 = (((long long)c.b * a.in.dims[0] + c.cell(0, 0)) * a.in.dims[1] + c.cell(1, 0)) * a.in.dims[2] + c.cell(2, 0);
  const long long slab = (long long)a.in.dims[1] * a.in.dims[2];
  for (int jx = 0; jx < c.cnt(0); ++jx) {
    unsigned long long w0[MY];
    unsigned nx[MY];
    int rk[MY], sh[MY];
#pragma unroll
    for (int jy = 0; jy < MY; ++jy) {
      const bool ok = jy < c.cnt(1);
      const long long cell = ok ? cell00 + jx * slab + (long long)jy * a.in.dims[2] : 0ll;
      const long long wi = cell >> 6;
      sh[jy] = ok ? (int)(cell & 63) : 64;
      w0[jy] = a.in.bits[wi];
      nx[jy] = bits32[2 * (wi < last_w ? wi + 1 : last_w)];          // (the last word has no successor: its field cannot straddle)
      rk[jy] = a.in.rank[wi];
    }
#pragma unroll
    for (int jy = 0; jy < MY; ++jy) {
      if (sh[jy] >= 64) continue;
      unsigned long long win = w0[jy] >> sh[jy];
      if (sh[jy]) win |= (unsigned long long)nx[jy] << (64 - sh[jy]);
      unsigned f = (unsigned)win & fmask;
      if (!f) continue;
      // row of the first hit: rank of the word + set bits below the field; later hits of the field follow consecutively
      // (the rank is cumulative across words, so a field that straddles two words needs nothing extra)
      int i = rk[jy] + __popcll(w0[jy] & ((1ull << sh[jy]) - 1ull));
      const int kxy = a.zk ? KS * (c.idx(1, jy) + KS * c.idx(0, jx)) : c.idx(0, jx) + KS * c.idx(1, jy);
      while (f) {
        const int t = __ffs((int)f) - 1;
        f &= f - 1;
        on_hit(i, kxy + kzs * c.idx(2, t));
        ++i;
      }
    }
  }
}

// fill pass: the pair ids of a block's 256 rows occupy one contiguous range of pair_ids (first[] is a prefix sum), so
// they are staged in LDS and written out coalesced.
template <int KS>
__global__ void __launch_bounds__(256) k_expand_csr_fill(ExpandCsrArgs a) {
  constexpr int CAP = 256 * 20;
  __shared__ int stage[CAP];
  const long long o0 = (long long)blockIdx.x * blockDim.x;
  const long long o = o0 + threadIdx.x;
  const long long o_end = min(a.n_out, o0 + 256);
  const int base = a.first[o0], total = a.first[o_end] - base;
  const bool staged = total <= CAP;
  if (a.d_total && blockIdx.x == 0 && threadIdx.x == 0) *a.d_total = a.first[a.n_out];
  if (o < a.n_out) {
    int wpos = a.first[o] - base;
    if (staged) csr_probe<KS>(a, o, [&](int i, int kidx) { stage[wpos++] = i * a.K + kidx; });
    else { wpos += base; csr_probe<KS>(a, o, [&](int i, int kidx) { a.pair_ids[wpos++] = i * a.K + kidx; }); }
  }
  if (!staged) return;
  __syncthreads();
  for (int t = threadIdx.x; t < total; t += 256) a.pair_ids[base + t] = stage[t];
}

// ---- one-pass pair lists in per-workgroup SLOTS (round 4) -------------------------------------------------------------------
// The count / scan / fill form probes every row twice (the probes are the cost: dependent bitmap + rank reads) because a row's
// position in pair_ids is a prefix sum over ALL rows before it.  Here a workgroup's 256 rows write into a slot of their own --
// pair_ids[w * SLOT ...), SLOT = 256 rows x the most pairs a row can have -- so positions need only a scan INSIDE the workgroup:
// probe once (hits parked in LDS), block scan of the counts, compact, write.  first[o] = absolute start of row o; a row's list
// ends where the next row's begins, except for the last row of a workgroup: wg_end[w].  Same lists, same order inside a row as
// the three-launch form (tests compare them row by row), so the gather-sum's results are bit-identical.  The buffer is sized for
// the worst case (no overflow path, nothing to re-run); only the written part costs bandwidth.
template <int KS>
__global__ void __launch_bounds__(256) k_expand_csr_slot(ExpandCsrArgs a, int slot, int* __restrict__ wg_end) {
  // (LDS budget = occupancy: the pass is latency-bound.  With a second 20 KB buffer to compact the workgroup's hits before the
  //  write it held 37 KB, four workgroups per CU, and took 560 us on the benchmark's last level -- as long as count + scan + fill.)
  constexpr int R = 12;                       // hits per row parked in LDS (rows with more re-probe: rare on surfaces)
  __shared__ int region[256 * R];
  __shared__ int wsum[4];
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  int cnt = 0;
  int* const mine = region + threadIdx.x;     // hit j of this thread at region[j * 256 + thread]: the lanes of a wave hit 64 different banks
  if (o < a.n_out) csr_probe<KS>(a, o, [&](int i, int kidx) { if (cnt < R) mine[cnt * 256] = i * a.K + kidx; ++cnt; });
  // exclusive scan of the counts over the workgroup
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int off = inc - cnt, total = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int sv = wsum[q]; if (q < w) off += sv; total += sv; }
  const int base = (int)blockIdx.x * slot;
  if (threadIdx.x == 0) {
    wg_end[blockIdx.x] = base + total;
    if (a.d_total) atomicAdd((unsigned long long*)a.d_total, (unsigned long long)total);   // (integer: order-independent)
  }
  if (o >= a.n_out) return;
  a.first[o] = base + off;
  // adjacent rows write adjacent runs: the wave's stores of one j cover ~64 * 4 consecutive positions, whole lines after a few j
  int* const dst = a.pair_ids + base + off;
  if (cnt <= R) { for (int j = 0; j < cnt; ++j) dst[j] = mine[j * 256]; }
  else { int wpos = 0; csr_probe<KS>(a, o, [&](int i, int kidx) { dst[wpos++] = i * a.K + kidx; }); }
}

__global__ void k_set_int(int* p, int v) { *p = v; }

// phase 1: output set + its grid index.  h_out: output lattice (pitch ts_out).  out_keys capacity >= min(n*K, cells).
extern "C" int pcc_coords_expand_grid(const int64_t* keys, int64_t n, int32_t kernel_size, const int32_t* h_out,
                                      uint64_t* bits, int32_t* rank, int64_t* out_keys, int64_t* d_count, void* ws,
                                      size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(keys && h_out && bits && rank && out_keys && d_count && n > 0, "pcc_coords_expand_grid: bad arguments");
  PCC_REQUIRE(kernel_size == 2 || kernel_size == 3 || kernel_size == 5, "pcc_coords_expand_grid: kernel_size %d unsupported", kernel_size);
  const int64_t words = pcc_grid_words(h_out);
  PCC_REQUIRE(words >= 1 && words < (1ll << 31), "pcc_coords_expand_grid: lattice too large (%lld words)", (long long)words);
  const int ts_out = h_out[6];
  PCC_REQUIRE(ts_out >= 1 && (ts_out & (ts_out - 1)) == 0, "pcc_coords_expand_grid: pitch must be a power of two");
  if (ws_bytes < pcc_grid_ws_bytes(words)) { pcc_set_error("pcc_coords_expand_grid: workspace too small"); return PCC_EWS; }
  PCC_CHECK_HIP(hipMemsetAsync(bits, 0, (size_t)words * 8, s));
  const dim3 g((unsigned)pcc_cdiv(n, 256), (unsigned)(kernel_size * kernel_size));
  unsigned long long* b = (unsigned long long*)bits;
  const int tsl = ilog2(ts_out);
  if (kernel_size == 2) k_expand_mark<2><<<g, 256, 0, s>>>(keys, n, ts_out, h_out[0], h_out[1], h_out[2], h_out[3], h_out[4], h_out[5], tsl, b);
  else if (kernel_size == 3) k_expand_mark<3><<<g, 256, 0, s>>>(keys, n, ts_out, h_out[0], h_out[1], h_out[2], h_out[3], h_out[4], h_out[5], tsl, b);
  else k_expand_mark<5><<<g, 256, 0, s>>>(keys, n, ts_out, h_out[0], h_out[1], h_out[2], h_out[3], h_out[4], h_out[5], tsl, b);
  PCC_LAUNCH_CHECK();
  return grid_rank(b, words, rank, h_out, tsl, out_keys, d_count, ws, ws_bytes, s);
}

extern "C" size_t pcc_expand_grid_csr_ws_bytes(int64_t n_out) { return pcc_scan_ws_bytes(n_out + 1) + 256; }

// phase 2 (n_out known to the host): CSR pair lists of the transposed map: first[n_out+1], pair_ids[n_in*K]
static int expand_grid_csr(const int64_t* out_keys, int64_t n_out, int32_t kernel_size, int32_t ts_out,
                           const uint64_t* in_bits, const int32_t* in_rank, const int32_t* h_in,
                           int64_t n_in, int32_t* first, int32_t* pair_ids, int64_t* d_total, void* ws, size_t ws_bytes, int zk,
                           void* stream);
extern "C" int pcc_coords_expand_grid_csr(const int64_t* out_keys, int64_t n_out, int32_t kernel_size, int32_t ts_out,
                                          const uint64_t* in_bits, const int32_t* in_rank, const int32_t* h_in,
                                          int64_t n_in, int32_t* first, int32_t* pair_ids, int64_t* d_total, void* ws,
                                          size_t ws_bytes, void* stream) {
  return expand_grid_csr(out_keys, n_out, kernel_size, ts_out, in_bits, in_rank, h_in, n_in, first, pair_ids, d_total, ws, ws_bytes, 0, stream);
}
// the same lists with the kernel offsets of the pair ids numbered z fastest (iz + KS*iy + KS*KS*ix): for a per-pair product
// buffer laid out [input row][kx][ky][kz][c], where the z-neighbours of a canonical (z-fastest) run of output rows are adjacent
extern "C" int pcc_coords_expand_grid_csr_zk(const int64_t* out_keys, int64_t n_out, int32_t kernel_size, int32_t ts_out,
                                             const uint64_t* in_bits, const int32_t* in_rank, const int32_t* h_in,
                                             int64_t n_in, int32_t* first, int32_t* pair_ids, int64_t* d_total, void* ws,
                                             size_t ws_bytes, void* stream) {
  return expand_grid_csr(out_keys, n_out, kernel_size, ts_out, in_bits, in_rank, h_in, n_in, first, pair_ids, d_total, ws, ws_bytes, 1, stream);
}
static int expand_grid_csr(const int64_t* out_keys, int64_t n_out, int32_t kernel_size, int32_t ts_out,
                           const uint64_t* in_bits, const int32_t* in_rank, const int32_t* h_in,
                           int64_t n_in, int32_t* first, int32_t* pair_ids, int64_t* d_total, void* ws, size_t ws_bytes, int zk,
                           void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(out_keys && in_bits && in_rank && h_in && first && pair_ids && n_out > 0, "pcc_coords_expand_grid_csr: bad arguments");
  PCC_REQUIRE(kernel_size == 2 || kernel_size == 3 || kernel_size == 5 || kernel_size == 7,
              "pcc_coords_expand_grid_csr: kernel_size %d unsupported", kernel_size);
  const int K = kernel_size * kernel_size * kernel_size;
  PCC_REQUIRE(n_in * K < (1ll << 31), "pcc_coords_expand_grid_csr: too many pairs");
  if (ws_bytes < pcc_expand_grid_csr_ws_bytes(n_out)) { pcc_set_error("pcc_coords_expand_grid_csr: workspace too small"); return PCC_EWS; }
  ExpandCsrArgs a;
  a.out_keys = out_keys; a.n_out = n_out; a.ts_out = ts_out; a.K = K; a.first = first; a.pair_ids = pair_ids; a.zk = zk;
  a.d_total = (long long*)d_total;
  a.in.bits = (const unsigned long long*)in_bits; a.in.rank = in_rank;
  for (int i = 0; i < 3; ++i) { a.in.lo[i] = h_in[i]; a.in.dims[i] = h_in[3 + i]; }
  a.in.ts_log2 = ilog2(h_in[6]); a.in.nbatch = h_in[7];
  const unsigned g = (unsigned)pcc_cdiv(n_out, 256);
  // pitch ratio: >= 2 leaves at most (KS + 1) / 2 compatible offsets per axis; 1 (stride-1 generative transpose) all KS
  PCC_REQUIRE(ts_out >= 1 && h_in[6] >= ts_out && h_in[6] % ts_out == 0, "pcc_coords_expand_grid_csr: input pitch %d is not a multiple of the output pitch %d", h_in[6], ts_out);
  const bool same_pitch = h_in[6] == ts_out;
  PCC_REQUIRE(!(same_pitch && kernel_size == 7), "pcc_coords_expand_grid_csr: 7-wide lists need an input pitch of at least twice the output pitch");
  if (kernel_size == 2) { if (same_pitch) k_expand_csr_count<2, 2><<<g, 256, 0, s>>>(a); else k_expand_csr_count<2, 1><<<g, 256, 0, s>>>(a); }
  else if (kernel_size == 3) { if (same_pitch) k_expand_csr_count<3, 3><<<g, 256, 0, s>>>(a); else k_expand_csr_count<3, 2><<<g, 256, 0, s>>>(a); }
  else if (kernel_size == 5) { if (same_pitch) k_expand_csr_count<5, 5><<<g, 256, 0, s>>>(a); else k_expand_csr_count<5, 3><<<g, 256, 0, s>>>(a); }
  else k_expand_csr_count<7, 4><<<g, 256, 0, s>>>(a);
  PCC_LAUNCH_CHECK();
#define PCC_EXPAND_CSR(KERNEL)                                                      \
  do {                                                                              \
    if (kernel_size == 2) KERNEL<2><<<g, 256, 0, s>>>(a);                           \
    else if (kernel_size == 3) KERNEL<3><<<g, 256, 0, s>>>(a);                      \
    else if (kernel_size == 5) KERNEL<5><<<g, 256, 0, s>>>(a);                      \
    else KERNEL<7><<<g, 256, 0, s>>>(a);                                            \
    PCC_LAUNCH_CHECK();                                                             \
  } while (0)
  // out_keys need not be the full expansion (any subset of rows, or a wider kernel restricted to a given set), so the
  // pair total is whatever the counts add up to: scan n_out + 1 entries
  PCC_TRY(pcc_scan_exclusive_i32(first, first, n_out + 1, ws, ws_bytes, s));
  PCC_EXPAND_CSR(k_expand_csr_fill);
#undef PCC_EXPAND_CSR
  return PCC_OK;
}

// one-pass slotted lists (k_expand_csr_slot): kernel sizes 5 and 7 at a pitch ratio >= 2 (the generative convolutions of the codec)
static int csr_slot_rows(int kernel_size) { const int m = (kernel_size + 1) / 2; return m * m * m; }
extern "C" int64_t pcc_expand_grid_csr_slot_elems(int64_t n_out, int32_t kernel_size) {
  if (kernel_size != 5 && kernel_size != 7) return -1;
  return pcc_cdiv(n_out > 0 ? n_out : 1, 256) * 256 * csr_slot_rows(kernel_size);
}
extern "C" int pcc_coords_expand_grid_csr_slots(const int64_t* out_keys, int64_t n_out, int32_t kernel_size, int32_t ts_out,
                                                const uint64_t* in_bits, const int32_t* in_rank, const int32_t* h_in, int64_t n_in,
                                                int32_t* first, int32_t* pair_ids, int32_t* wg_end, int64_t* d_total, int32_t zk,
                                                void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(out_keys && in_bits && in_rank && h_in && first && pair_ids && wg_end && n_out > 0, "pcc_coords_expand_grid_csr_slots: bad arguments");
  PCC_REQUIRE(kernel_size == 5 || kernel_size == 7, "pcc_coords_expand_grid_csr_slots: kernel_size %d unsupported", kernel_size);
  const int K = kernel_size * kernel_size * kernel_size;
  PCC_REQUIRE(n_in * K < (1ll << 31), "pcc_coords_expand_grid_csr_slots: too many pairs");
  PCC_REQUIRE(pcc_expand_grid_csr_slot_elems(n_out, kernel_size) < (1ll << 31), "pcc_coords_expand_grid_csr_slots: too many rows for 32-bit list positions");
  PCC_REQUIRE(ts_out >= 1 && h_in[6] >= 2 * ts_out && h_in[6] % ts_out == 0,
              "pcc_coords_expand_grid_csr_slots: the input pitch %d must be at least twice the output pitch %d", h_in[6], ts_out);
  ExpandCsrArgs a;
  a.out_keys = out_keys; a.n_out = n_out; a.ts_out = ts_out; a.K = K; a.first = first; a.pair_ids = pair_ids; a.zk = zk;
  a.d_total = (long long*)d_total;
  a.in.bits = (const unsigned long long*)in_bits; a.in.rank = in_rank;
  for (int i = 0; i < 3; ++i) { a.in.lo[i] = h_in[i]; a.in.dims[i] = h_in[3 + i]; }
  a.in.ts_log2 = ilog2(h_in[6]); a.in.nbatch = h_in[7];
  const unsigned g = (unsigned)pcc_cdiv(n_out, 256);
  const int slot = 256 * csr_slot_rows(kernel_size);
  if (d_total) PCC_CHECK_HIP(hipMemsetAsync(d_total, 0, sizeof(int64_t), s));
  if (kernel_size == 5) k_expand_csr_slot<5><<<g, 256, 0, s>>>(a, slot, wg_end);
  else k_expand_csr_slot<7><<<g, 256, 0, s>>>(a, slot, wg_end);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// conv-form kernel map (one segment, nbr[k][o] = input row or -1) from CSR pair lists: lets the pair-list convolution
// evaluate a transposed conv on a SUBSET of its output rows (the rows that survive pruning) without the dense T
__global__ void k_csr_to_nbr(const int* __restrict__ first, const int* __restrict__ pair_ids, long long n_out, int K,
                             int* __restrict__ nbr) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_out) return;
  for (int t = first[o]; t < first[o + 1]; ++t) {
    const int pid = pair_ids[t];
    const int i = pid / K, k = pid - i * K;
    nbr[(long long)k * n_out + o] = i;
  }
}

extern "C" int pcc_map_from_csr(const int32_t* first, const int32_t* pair_ids, int64_t n_out, int32_t kernel_size,
                                int32_t* hdr, int32_t* nbr, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(first && pair_ids && hdr && nbr && n_out > 0, "pcc_map_from_csr: bad arguments");
  const int K = kernel_size * kernel_size * kernel_size;
  PCC_REQUIRE(K >= 1 && K <= 192 && (int64_t)K * n_out < (1ll << 31), "pcc_map_from_csr: K=%d unsupported", K);
  SegPlan plan;
  memset(&plan, 0, sizeof(plan));
  plan.K = K; plan.nseg = 1; plan.listed = 0; plan.k_count[0] = K; plan.koff_begin[0] = 0;
  plan_order(plan, kernel_size);
  k_write_hdr<<<1, 192, 0, s>>>(plan, nullptr, 0, n_out, hdr);
  PCC_LAUNCH_CHECK();
  PCC_CHECK_HIP(hipMemsetAsync(nbr, 0xFF, (size_t)K * n_out * sizeof(int), s));
  k_csr_to_nbr<<<(unsigned)pcc_cdiv(n_out, 256), 256, 0, s>>>(first, pair_ids, n_out, K, nbr);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// dense [K][n_out] view (tests / inspection)
__global__ void k_map_dense(const int* __restrict__ hdr, const int* __restrict__ nbr, const int* __restrict__ rows,
                            int64_t n_out, int* __restrict__ dense) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (p >= n_out) return;
  const int nseg = hdr[HDR_NSEG];
  int s = 0;
  for (; s < nseg - 1; ++s) {
    const int* sg = hdr + HDR_SEG0 + s * SEG_WORDS;
    if (p < (int64_t)sg[SEG_POS_BEGIN] + sg[SEG_POS_COUNT]) break;
  }
  const int* sg = hdr + HDR_SEG0 + s * SEG_WORDS;
  if (j >= sg[SEG_K_COUNT]) return;
  const int kid = hdr[HDR_KOFFS + sg[SEG_KOFF_BEGIN] + j];
  const int64_t nb = ((int64_t)(unsigned)sg[SEG_NBR_LO]) | ((int64_t)sg[SEG_NBR_HI] << 32);
  const int o = rows ? rows[p] : (int)p;
  dense[(int64_t)kid * n_out + o] = nbr[nb + (int64_t)j * sg[SEG_POS_COUNT] + (p - sg[SEG_POS_BEGIN])];
}

extern "C" int pcc_map_to_dense(const int32_t* hdr, const int32_t* nbr, const int32_t* rows, int64_t n_out,
                                int32_t K, int32_t* dense, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_out <= 0) return PCC_OK;
  PCC_REQUIRE(hdr && nbr && dense && K >= 1 && K <= 192, "pcc_map_to_dense: bad arguments");
  PCC_CHECK_HIP(hipMemsetAsync(dense, 0xFF, (size_t)n_out * K * sizeof(int), s));
  dim3 grid((unsigned)pcc_cdiv(n_out, 256), (unsigned)K);
  k_map_dense<<<grid, 256, 0, s>>>(hdr, nbr, rows, n_out, dense);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}
