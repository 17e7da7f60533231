// Sparse convolution forward for gfx950: output-stationary implicit GEMM on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32), deterministic (no atomics, fixed summation order).
//
// A workgroup owns BM consecutive positions of one map segment and BN output channels.  The
// reduction dimension is the flattened (active kernel offset, input channel) axis, consumed in
// chunks of 32: the BM gathered feature-row pieces and the BN weight rows of a chunk are staged
// through LDS as [row][32+4] tiles (16-byte pad: conflict-free ds_read_b128 / ds_write_b128), and
// every wave multiplies its 32x32 tiles with 4 MFMAs per pair of 16-byte fragment reads.
// Kernel offsets for which no position of the tile has a neighbour are skipped (wave ballots over
// the neighbour table), so sparse tiles do not pay for empty offsets.
//
// The same kernel body computes the fused GDN / IGDN (model/blocks.py:38-57): A = |x|, W = gamma^T,
// epilogue x / (acc + beta) or x * (acc + beta).
//
// Thin outputs (Cout <= 4: occupancy logits, colours, model/transforms.py:141-160) are gather-bound;
// they use a VALU kernel with lanes spread over the input channels of a row.
#include <vector>

#include "pcc_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int LDS_LD = 36;   // floats per LDS tile row: 32 + 4 pad
static constexpr int MAXK = 128;    // kernel offsets per segment (K <= 125)
static constexpr int MAXK_T = 512;  // offsets of the input-stationary transposed conv (a flat GEMM: 7^3 composites fit)

enum { MODE_CONV = 0, MODE_GDN = 1, MODE_IGDN = 2 };

struct ConvArgs {
  const float* feat;      // [n_in, cin]
  const float* wp;        // packed weights [K*ppo][cout_pad][CB]
  const float* bias;      // [cout] or null (GDN: beta_eff)
  const int* hdr;         // map header (null: identity, one segment of n_out positions)
  const int* nbr;
  const int* rows;
  float* out;             // [n_out, cout]
  long long n_out;
  long long n_in;         // rows of feat (buffer-addressed gathers)
  long long wp_elems;     // floats in wp
  const int* pair_in = nullptr;   // pair mode (pcc_conv_fwd_pairs): input row of every (padded) pair, -1 = padding
  const int* tile_k = nullptr;    // pair mode: kernel offset of each 128-pair tile
  const long long* n_tiles = nullptr;   // pair mode: device count of tiles (the grid is an upper bound)
  int cin, cout, cout_pad;
  int cb_log2;            // log2(CB), CB = min(cin, 32)
  int ppo;                // pieces per offset = cin / CB
  int act;
  float slope;
};

__host__ __device__ inline int bn_for(int cout) { return cout >= 128 ? 128 : (cout > 32 ? 64 : 32); }

// BUF: feature rows and weight rows are fetched with buffer loads whose offset is out of range for an absent neighbour
// (reads 0, no memory access): no per-row branch, no zero fill, 32-bit address arithmetic and a fixed number of loads
// in flight, so the s_waitcnt distances the compiler derives are exact.  Needs feat and wp below 4 GB each.
static constexpr unsigned BUF_OOB = 0xFFFF0000u;
static constexpr long long BUF_MAX_BYTES = 0xFFFE0000ll;

template <int WM, int WN, int TM, int TN, int MODE, bool BUF>
__global__ void __launch_bounds__(256) k_conv_mfma(ConvArgs a) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];
  __shared__ unsigned char act_flag[MAXK];
  __shared__ unsigned char act_list[MAXK];   // segment-local offset slot
  __shared__ unsigned char act_kid[MAXK];    // kernel offset id (weight index)
  __shared__ int s_nact;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

  // ---- XCD-aware work mapping --------------------------------------------------------------------
  // Workgroups are dealt round-robin over the 8 XCDs (private L2 each).  Neighbouring tiles gather almost the same
  // input rows, so XCD x is given a CONTIGUOUS range of work ids: its L2 then serves the re-reads that otherwise go
  // to the fabric 8 times (measured with rocprofv3 FETCH_SIZE: 12-25x the compulsory bytes without this).  The
  // column blocks of one row tile are adjacent ids (same gathered rows).  Speed only, never correctness.
  const int cpx = gridDim.x >> 3;                         // grid is a multiple of 8
  const int wid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
  const int gy = a.cout_pad / BN;
  const int tile_id = wid / gy;
  const int colblock = (wid - tile_id * gy) * BN;

  // ---- locate (segment, tile) --------------------------------------------------------------
  int pos0, npos, k_count, koff_begin;
  long long seg_pos_count;
  const int* seg_nbr = nullptr;
  const bool pair_mode = (a.pair_in != nullptr);
  bool identity = (a.hdr == nullptr) && !pair_mode;
  if (pair_mode) {          // one kernel offset per tile, rows = compacted pairs of that offset
    if (tile_id >= *a.n_tiles) return;
    pos0 = tile_id * BM; npos = BM; k_count = 1; koff_begin = 0; seg_pos_count = 0;
    seg_nbr = a.pair_in + pos0;
  } else if (identity) {
    const long long p0 = (long long)tile_id * BM;
    if (p0 >= a.n_out) return;
    pos0 = (int)p0;
    npos = (int)min((long long)BM, a.n_out - p0);
    k_count = 1; koff_begin = 0; seg_pos_count = a.n_out;
  } else {
    const int nseg = a.hdr[HDR_NSEG];
    int tile = tile_id, s = 0;
    bool found = false;
    int pb = 0, pc = 0;
    for (; s < nseg; ++s) {
      const int* sg = a.hdr + HDR_SEG0 + s * SEG_WORDS;
      pb = sg[SEG_POS_BEGIN]; pc = sg[SEG_POS_COUNT];
      const int tiles = (pc + BM - 1) / BM;
      if (tile < tiles) { found = true; break; }
      tile -= tiles;
    }
    if (!found) return;   // grid is an upper bound on the tile count
    const int* sg = a.hdr + HDR_SEG0 + s * SEG_WORDS;
    k_count = sg[SEG_K_COUNT];
    koff_begin = sg[SEG_KOFF_BEGIN];
    const long long nb = ((long long)(unsigned)sg[SEG_NBR_LO]) | ((long long)sg[SEG_NBR_HI] << 32);
    seg_nbr = a.nbr + nb;
    seg_pos_count = pc;
    const int local0 = tile * BM;
    pos0 = pb + local0;
    npos = min(BM, pc - local0);
    seg_nbr += local0;     // seg_nbr[j * seg_pos_count + r] = input row of tile row r for slot j
  }

  // ---- active offsets of this tile ---------------------------------------------------------
  if (pair_mode) {
    if (tid == 0) { act_list[0] = 0; act_kid[0] = (unsigned char)a.tile_k[tile_id]; s_nact = 1; }
  } else if (identity) {
    if (tid == 0) { act_list[0] = 0; act_kid[0] = 0; s_nact = 1; }
  } else {
    for (int j = w; j < k_count; j += 4) {
      bool any = false;
      for (int r = lane; r < npos; r += 64) any |= (seg_nbr[(long long)j * seg_pos_count + r] >= 0);
      const unsigned long long m = __ballot(any);
      if (lane == 0) act_flag[j] = m ? 1 : 0;
    }
    __syncthreads();
    if (w == 0) {
      int n = 0;
      for (int j0 = 0; j0 < k_count; j0 += 64) {
        const int u = j0 + lane;                                          // visiting position -> slot
        const int j = (u < k_count) ? a.hdr[HDR_ORDER + koff_begin + u] : 0;
        const bool f = (u < k_count) && act_flag[j];
        const unsigned long long m = __ballot(f);
        if (f) {
          const int p = n + __popcll(m & ((1ull << lane) - 1ull));
          act_list[p] = (unsigned char)j;
          act_kid[p] = (unsigned char)a.hdr[HDR_KOFFS + koff_begin + j];
        }
        n += __popcll(m);
      }
      if (lane == 0) s_nact = n;
    }
  }
  __syncthreads();
  const int nact = s_nact;

  const int CB = 1 << a.cb_log2;
  const int ppc_log2 = 5 - a.cb_log2;                 // pieces per 32-wide chunk
  const int npieces = nact * a.ppo;
  const int nchunks = (npieces + (1 << ppc_log2) - 1) >> ppc_log2;

  // staging role of this thread: 16-byte part `part` of tile rows r0 + 32*i
  const int part = tid & 7;
  const int r0 = tid >> 3;
  const int kk0 = part * 4;
  const int piece_in_chunk = kk0 >> a.cb_log2;
  const int within = kk0 & (CB - 1);
  constexpr int AI = BM / 32, BI = BN / 32;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm = w / WN, wn = w % WN;
  const int half = lane >> 5, r31 = lane & 31;

  // ---- software pipeline over the 32-wide chunks -------------------------------------------------
  //   neighbour rows of chunk c+2  -> registers   (dependent-load chain hidden two chunks ahead)
  //   global loads  of chunk c+1  -> registers   (in flight while chunk c is multiplied)
  //   chunk c: registers -> LDS -> fragments -> MFMA
  auto chunk_ids = [&](int c, int& ai, int& cbi, bool& pvalid) {
    const int piece = (c << ppc_log2) + piece_in_chunk;
    ai = piece / a.ppo;            // active-offset index of my 16-byte part
    cbi = piece - ai * a.ppo;      // channel block within the offset
    pvalid = ai < nact;
  };
  __amdgpu_buffer_rsrc_t rsA, rsB;
  if constexpr (BUF) {
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.feat), (short)0,
                                            (int)(unsigned)((size_t)a.n_in * a.cin * 4), 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), (short)0, (int)(unsigned)((size_t)a.wp_elems * 4),
                                            0x00020000);
  }
  const unsigned cin_bytes = (unsigned)a.cin * 4u;
  auto load_rows = [&](int ai, bool pvalid, int (&rows)[AI]) {
    const int slot = pvalid ? act_list[ai] : 0;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int r = r0 + 32 * i;
      if constexpr (BUF) {         // tail rows repeat the tile's last row (never stored); !pvalid is handled in issue()
        const int rc = min(r, npos - 1);
        rows[i] = identity ? (pos0 + rc) : seg_nbr[(long long)slot * seg_pos_count + rc];
      } else {
        int v = -1;
        if (pvalid && r < npos) v = identity ? (pos0 + r) : seg_nbr[(long long)slot * seg_pos_count + r];
        rows[i] = v;
      }
    }
  };
  auto issue = [&](int ai, int cbi, bool pvalid, const int (&rows)[AI], float4 (&av)[AI], float4 (&bv)[BI]) {
    if constexpr (BUF) {
      const unsigned cb_off = (unsigned)(((cbi << a.cb_log2) + within) * 4);
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const unsigned off = (rows[i] >= 0 && pvalid) ? (unsigned)rows[i] * cin_bytes + cb_off : BUF_OOB;
        av[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
      }
      const unsigned wbase = pvalid ? (unsigned)((act_kid[ai] * a.ppo + cbi) * a.cout_pad + colblock + r0) : 0u;
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const unsigned off = pvalid ? (((wbase + 32u * i) << a.cb_log2) + within) * 4u : BUF_OOB;
        bv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, off, 0, 0));
      }
    } else {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rows[i] >= 0)
          av[i] = *reinterpret_cast<const float4*>(a.feat + (long long)rows[i] * a.cin + (cbi << a.cb_log2) + within);
      }
      const long long wbase = pvalid ? ((long long)(act_kid[ai] * a.ppo + cbi) * a.cout_pad) : 0;
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pvalid) {
          const int col = colblock + r0 + 32 * i;
          bv[i] = *reinterpret_cast<const float4*>(a.wp + ((wbase + col) << a.cb_log2) + within);
        }
      }
    }
  };

  int rows_cur[AI], rows_nxt[AI];
  float4 av[AI], bv[BI];
  int ai_c, cbi_c, ai_n = -1, cbi_n = 0;
  bool pv_c, pv_n = false;
  if (nchunks > 0) {
    chunk_ids(0, ai_c, cbi_c, pv_c);
    load_rows(ai_c, pv_c, rows_cur);
    issue(ai_c, cbi_c, pv_c, rows_cur, av, bv);
    if (nchunks > 1) {
      chunk_ids(1, ai_n, cbi_n, pv_n);
      if (ai_n != ai_c) load_rows(ai_n, pv_n, rows_nxt);
      else {
#pragma unroll
        for (int i = 0; i < AI; ++i) rows_nxt[i] = rows_cur[i];
      }
    }
  }

  for (int c = 0; c < nchunks; ++c) {
    if (MODE != MODE_CONV) {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        av[i].x = fabsf(av[i].x); av[i].y = fabsf(av[i].y); av[i].z = fabsf(av[i].z); av[i].w = fabsf(av[i].w);
      }
    }
    __syncthreads();   // previous chunk's fragment reads are done
#pragma unroll
    for (int i = 0; i < AI; ++i)
      *reinterpret_cast<float4*>(&As[(r0 + 32 * i) * LDS_LD + kk0]) = av[i];
#pragma unroll
    for (int i = 0; i < BI; ++i)
      *reinterpret_cast<float4*>(&Bs[(r0 + 32 * i) * LDS_LD + kk0]) = bv[i];
    __syncthreads();
    if (c + 1 < nchunks) {          // next chunk's global loads fly during this chunk's MFMAs
#pragma unroll
      for (int i = 0; i < AI; ++i) rows_cur[i] = rows_nxt[i];
      ai_c = ai_n; cbi_c = cbi_n; pv_c = pv_n;
      issue(ai_c, cbi_c, pv_c, rows_cur, av, bv);
      if (c + 2 < nchunks) {
        chunk_ids(c + 2, ai_n, cbi_n, pv_n);
        if (ai_n != ai_c) load_rows(ai_n, pv_n, rows_nxt);
      }
    }
    if constexpr (BUF) __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs, not next to its use
    // LDS -> fragments -> MFMA
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const float4*>(&As[((wm * TM + i) * 32 + r31) * LDS_LD + g * 8 + half * 4]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *reinterpret_cast<const float4*>(&Bs[((wn * TN + j) * 32 + r31) * LDS_LD + g * 8 + half * 4]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
  }

  // ---- epilogue: bias, activation (or GDN), store -------------------------------------------
  // Full tiles written to consecutive rows take a branch-free path: one base pointer per lane, the activation chosen
  // once per tile.  (The general loop below costs ~50 instructions per element -- row-list lookups, tail checks and
  // the activation switch for each of the 64 values a lane holds -- which is as much as the whole MFMA phase of a
  // 128-deep GEMM tile.)
  if (!a.rows && npos == BM) {
    const size_t lane_off = (size_t)(pos0 + wm * TM * 32 + 4 * half) * a.cout + colblock + wn * TN * 32 + r31;
    float* const lane_out = a.out + lane_off;
    const float* const lane_x = a.feat + lane_off;             // GDN / IGDN: cin == cout, same element of the input
    auto store_tile = [&](auto actf) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = colblock + (wn * TN + j) * 32 + r31;
        if (col >= a.cout) continue;
        const float b = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const size_t o = (size_t)(i * 32 + (e & 3) + 8 * (e >> 2)) * a.cout + j * 32;
            lane_out[o] = actf(acc[i][j][e] + b, o);
          }
      }
    };
    if (MODE == MODE_GDN) store_tile([&](float v, size_t o) { return lane_x[o] / v; });
    else if (MODE == MODE_IGDN) store_tile([&](float v, size_t o) { return lane_x[o] * v; });
    else if (a.act == PCC_ACT_RELU) store_tile([](float v, size_t) { return fmaxf(v, 0.f); });
    else if (a.act == PCC_ACT_LEAKY) { const float sl = a.slope; store_tile([sl](float v, size_t) { return v >= 0.f ? v : v * sl; }); }
    else store_tile([](float v, size_t) { return v; });
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = colblock + (wn * TN + j) * 32 + r31;
    if (col >= a.cout) continue;
    const float b = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (r >= npos) continue;
        const long long orow = a.rows ? a.rows[pos0 + r] : (pos0 + r);
        float v = acc[i][j][e] + b;
        if (MODE == MODE_CONV) {
          if (a.act == PCC_ACT_RELU) v = fmaxf(v, 0.f);
          else if (a.act == PCC_ACT_LEAKY) v = v >= 0.f ? v : v * a.slope;
        } else {
          const float x = a.feat[orow * a.cin + col];
          v = (MODE == MODE_GDN) ? x / v : x * v;
        }
        a.out[orow * a.cout + col] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// thin outputs (cout <= 4) or channel counts the MFMA tiling does not take: VALU, gather-bound.
// Wt layout [K][cout][cin].  LPR lanes share one output position.
// ------------------------------------------------------------------------------------------
struct ThinArgs {
  const float* feat; const float* wt; const float* bias;
  const int* hdr; const int* nbr; const int* rows;
  float* out; long long n_out; int cin, cout, act; float slope; int lpr_log2;
};

template <int VEC> struct ThinVec;
template <> struct ThinVec<4> { typedef float4 T; };
template <> struct ThinVec<1> { typedef float T; };
__device__ inline float thin_dot(float4 x, float4 w) { return x.x * w.x + x.y * w.y + x.z * w.z + x.w * w.w; }
__device__ inline float thin_dot(float x, float w) { return x * w; }
__device__ inline void thin_zero(float4& v) { v = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ inline void thin_zero(float& v) { v = 0.f; }
__device__ inline void thin_acc(float4& a, const float4 x) { a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; }
__device__ inline void thin_acc(float& a, const float x) { a += x; }
__device__ inline float act1(float v, int act, float slope) {
  if (act == PCC_ACT_RELU) return fmaxf(v, 0.f);
  if (act == PCC_ACT_LEAKY) return v >= 0.f ? v : v * slope;
  return v;
}
__device__ inline void thin_act(float4& a, int act, float s) { a.x = act1(a.x, act, s); a.y = act1(a.y, act, s); a.z = act1(a.z, act, s); a.w = act1(a.w, act, s); }
__device__ inline void thin_act(float& a, int act, float s) { a = act1(a, act, s); }

// LPR lanes share one output position, each lane owns VEC consecutive input channels per pass.  Offsets are
// processed in batches of JB with all neighbour-index loads, then all feature loads, issued back to back
// (memory-level parallelism instead of a dependent chain per offset).
template <int COUT_MAX, int VEC>
__global__ void __launch_bounds__(256) k_conv_thin(ThinArgs a) {
  typedef typename ThinVec<VEC>::T VT;
  constexpr int JB = 9;
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << a.lpr_log2;
  const int rpw = 64 >> a.lpr_log2;                       // rows per wave
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long p = wave * rpw + (lane >> a.lpr_log2);  // position handled by my lane group
  const int cl = lane & (lpr - 1);
  const bool valid = p < a.n_out;
  const int cvec = a.cin / VEC;                           // vectors per row

  int k_count = 1, koff_begin = 0;
  long long seg_pos_count = a.n_out, local = p;
  const int* seg_nbr = nullptr;
  const bool identity = (a.hdr == nullptr);
  if (!identity && valid) {
    const int nseg = a.hdr[HDR_NSEG];
    int s = 0;
    for (; s < nseg - 1; ++s) {
      const int* sg = a.hdr + HDR_SEG0 + s * SEG_WORDS;
      if (p < (long long)sg[SEG_POS_BEGIN] + sg[SEG_POS_COUNT]) break;
    }
    const int* sg = a.hdr + HDR_SEG0 + s * SEG_WORDS;
    k_count = sg[SEG_K_COUNT];
    koff_begin = sg[SEG_KOFF_BEGIN];
    seg_pos_count = sg[SEG_POS_COUNT];
    local = p - sg[SEG_POS_BEGIN];
    seg_nbr = a.nbr + (((long long)(unsigned)sg[SEG_NBR_LO]) | ((long long)sg[SEG_NBR_HI] << 32));
  }
  float acc[COUT_MAX];
#pragma unroll
  for (int o = 0; o < COUT_MAX; ++o) acc[o] = 0.f;
  if (valid) {
    for (int cv = cl; cv < cvec; cv += lpr) {
      for (int j0 = 0; j0 < k_count; j0 += JB) {
        int ir[JB];
#pragma unroll
        for (int u = 0; u < JB; ++u) {
          const int j = j0 + u;
          ir[u] = (j < k_count) ? (identity ? (int)p : seg_nbr[(long long)j * seg_pos_count + local]) : -1;
        }
        VT x[JB];
#pragma unroll
        for (int u = 0; u < JB; ++u) {
          thin_zero(x[u]);
          if (ir[u] >= 0) x[u] = reinterpret_cast<const VT*>(a.feat + (long long)ir[u] * a.cin)[cv];
        }
#pragma unroll
        for (int u = 0; u < JB; ++u) {
          const int j = j0 + u;
          if (j < k_count) {
            const int kid = identity ? 0 : a.hdr[HDR_KOFFS + koff_begin + j];
            const float* wk = a.wt + (long long)kid * a.cout * a.cin;
#pragma unroll
            for (int o = 0; o < COUT_MAX; ++o)
              if (o < a.cout) acc[o] += thin_dot(x[u], reinterpret_cast<const VT*>(wk + o * a.cin)[cv]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < COUT_MAX; ++o)
    for (int d = lpr >> 1; d >= 1; d >>= 1) acc[o] += __shfl_xor(acc[o], d);
  if (valid && cl == 0) {
    const long long orow = a.rows ? a.rows[p] : p;
#pragma unroll
    for (int o = 0; o < COUT_MAX; ++o) {
      if (o >= a.cout) break;
      float v = acc[o] + (a.bias ? a.bias[o] : 0.f);
      if (a.act == PCC_ACT_RELU) v = fmaxf(v, 0.f);
      else if (a.act == PCC_ACT_LEAKY) v = v >= 0.f ? v : v * a.slope;
      a.out[orow * a.cout + o] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Thin outputs, two-pass form (cout <= 4, cin <= 64):  out[o] = b + sum_k  <feat[nbr_k(o)], w_k>
//   pass 1  t[k*cout+co][i] = <feat[i], w_k[co]>      per input row, features read ONCE, coalesced writes
//   pass 2  out[o][co]      = b + sum_k t[k*cout+co][nbr_k(o)]   scalar gathers, near-contiguous per offset
// 16x (cin=16) to 64x (cin=64) fewer gathered bytes than fetching whole neighbour rows per offset.
// ------------------------------------------------------------------------------------------
template <int CIN>
__global__ void __launch_bounds__(256) k_thin_project(const float* __restrict__ feat, long long n_in,
                                                      const float* __restrict__ wt, int kc, float* __restrict__ t) {
  extern __shared__ __attribute__((aligned(16))) float w_s[];
  for (int i = threadIdx.x; i < kc * CIN; i += 256) w_s[i] = wt[i];
  __syncthreads();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_in) return;
  float4 x[CIN / 4];
#pragma unroll
  for (int c = 0; c < CIN / 4; ++c) x[c] = reinterpret_cast<const float4*>(feat + i * CIN)[c];
  for (int k = 0; k < kc; ++k) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < CIN / 4; ++c) {
      const float4 w = reinterpret_cast<const float4*>(w_s + k * CIN)[c];   // wave-uniform address: LDS broadcast
      acc += x[c].x * w.x + x[c].y * w.y + x[c].z * w.z + x[c].w * w.w;
    }
    t[(long long)k * n_in + i] = acc;
  }
}

struct ThinGatherArgs {
  const float* t; const float* bias; const int* hdr; const int* nbr; const int* rows;
  float* out; long long n_in, n_out; int cout, act; float slope;
};

template <int COUT_MAX>
__global__ void __launch_bounds__(256) k_thin_gather(ThinGatherArgs a) {
  constexpr int JB = 9;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n_out) return;
  int k_count = 1, koff_begin = 0;
  long long spc = a.n_out, local = p;
  const int* seg_nbr = nullptr;
  const bool identity = (a.hdr == nullptr);
  if (!identity) {
    const int nseg = a.hdr[HDR_NSEG];
    int s = 0;
    for (; s < nseg - 1; ++s) {
      const int* sg = a.hdr + HDR_SEG0 + s * SEG_WORDS;
      if (p < (long long)sg[SEG_POS_BEGIN] + sg[SEG_POS_COUNT]) break;
    }
    const int* sg = a.hdr + HDR_SEG0 + s * SEG_WORDS;
    k_count = sg[SEG_K_COUNT]; koff_begin = sg[SEG_KOFF_BEGIN]; spc = sg[SEG_POS_COUNT];
    local = p - sg[SEG_POS_BEGIN];
    seg_nbr = a.nbr + (((long long)(unsigned)sg[SEG_NBR_LO]) | ((long long)sg[SEG_NBR_HI] << 32));
  }
  float acc[COUT_MAX];
#pragma unroll
  for (int o = 0; o < COUT_MAX; ++o) acc[o] = 0.f;
  for (int j0 = 0; j0 < k_count; j0 += JB) {
    int ir[JB];
#pragma unroll
    for (int u = 0; u < JB; ++u)
      ir[u] = (j0 + u < k_count) ? (identity ? (int)p : seg_nbr[(long long)(j0 + u) * spc + local]) : -1;
    float v[JB][COUT_MAX];
#pragma unroll
    for (int u = 0; u < JB; ++u) {
      const int kid = (ir[u] >= 0 && !identity) ? a.hdr[HDR_KOFFS + koff_begin + j0 + u] : 0;
#pragma unroll
      for (int o = 0; o < COUT_MAX; ++o)
        v[u][o] = (ir[u] >= 0 && o < a.cout) ? a.t[(long long)(kid * a.cout + o) * a.n_in + ir[u]] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < JB; ++u)
#pragma unroll
      for (int o = 0; o < COUT_MAX; ++o) acc[o] += v[u][o];
  }
  const long long orow = a.rows ? a.rows[p] : p;
#pragma unroll
  for (int o = 0; o < COUT_MAX; ++o)
    if (o < a.cout) a.out[orow * a.cout + o] = act1(acc[o] + (a.bias ? a.bias[o] : 0.f), a.act, a.slope);
}

// ------------------------------------------------------------------------------------------
// Narrow outputs with weights that fit LDS (4 < cout <= 16, cin in {16,32,64}): wave-autonomous kernel on
// v_mfma_f32_16x16x4_f32.  All K weight slices sit in LDS for the whole (persistent) workgroup; each wave owns
// 32 positions (two 16-row MFMA tiles), reads the neighbour rows straight from global memory into the MFMA A
// layout (lane = row, 16-byte k-quads) and never meets a workgroup barrier in its main loop.  No padding to a
// 32-wide column tile, offsets with no neighbour in the wave's 32 rows are skipped by ballot.
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Wave16Args {
  const float* feat; const float* wl; const float* bias; const int* hdr; const int* nbr; const int* rows;
  float* out; long long n_out, n_in; int K, cout, act; float slope;
};

template <int CIN>
__global__ void __launch_bounds__(512) k_conv_wave16(Wave16Args a) {
  constexpr int LD = CIN + 4;
  constexpr int G = CIN / 16;
  constexpr int NW = 8;                                            // waves per workgroup
  extern __shared__ __attribute__((aligned(16))) float wl_s[];   // [K][16][LD]
  for (int i = threadIdx.x; i < a.K * 16 * (CIN / 4); i += 512) {
    const int row = i / (CIN / 4), c4 = i - row * (CIN / 4);
    reinterpret_cast<float4*>(wl_s + row * LD)[c4] = reinterpret_cast<const float4*>(a.wl + (long long)row * CIN)[c4];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
  const bool identity = (a.hdr == nullptr);
  const int nseg = identity ? 1 : a.hdr[HDR_NSEG];
  long long total_tiles = 0;
  if (identity) total_tiles = (a.n_out + 31) / 32;
  else
    for (int s = 0; s < nseg; ++s) total_tiles += (a.hdr[HDR_SEG0 + s * SEG_WORDS + SEG_POS_COUNT] + 31) / 32;

  // XCD x sweeps its own contiguous eighth of the tiles with all of its waves side by side (L2 locality of the gathers)
  const int cpx = gridDim.x >> 3;                                   // workgroups per XCD (grid is a multiple of 8)
  const long long per_xcd = (total_tiles + 7) / 8;
  const long long xcd_lo = (long long)(blockIdx.x & 7) * per_xcd;
  const long long xcd_hi = min(total_tiles, xcd_lo + per_xcd);
  for (long long wt = xcd_lo + (long long)(blockIdx.x >> 3) * NW + (threadIdx.x >> 6); wt < xcd_hi;
       wt += (long long)cpx * NW) {
    long long pos0, spc;
    int npos, k_count = 1, koff_begin = 0;
    const int* seg_nbr = nullptr;
    if (identity) {
      pos0 = wt * 32; npos = (int)min(32ll, a.n_out - pos0); spc = a.n_out;
    } else {
      long long tile = wt;
      int s = 0;
      for (; s < nseg - 1; ++s) {
        const long long tiles = (a.hdr[HDR_SEG0 + s * SEG_WORDS + SEG_POS_COUNT] + 31) / 32;
        if (tile < tiles) break;
        tile -= tiles;
      }
      const int* sg = a.hdr + HDR_SEG0 + s * SEG_WORDS;
      k_count = sg[SEG_K_COUNT]; koff_begin = sg[SEG_KOFF_BEGIN]; spc = sg[SEG_POS_COUNT];
      const long long local0 = tile * 32;
      pos0 = sg[SEG_POS_BEGIN] + local0;
      npos = (int)min(32ll, spc - local0);
      seg_nbr = a.nbr + (((long long)(unsigned)sg[SEG_NBR_LO]) | ((long long)sg[SEG_NBR_HI] << 32)) + local0;
    }
    const bool vA = r16 < npos, vB = 16 + r16 < npos;
    f32x4 accA0 = {0.f, 0.f, 0.f, 0.f}, accA1 = accA0, accB0 = accA0, accB1 = accA0;
    auto fetch = [&](int j, int& iA, int& iB) {     // natural slot order: no dependent table read ahead of the index load
      iA = -1; iB = -1;
      if (j < k_count) {
        if (vA) iA = identity ? (int)(pos0 + r16) : seg_nbr[(long long)j * spc + r16];
        if (vB) iB = identity ? (int)(pos0 + 16 + r16) : seg_nbr[(long long)j * spc + 16 + r16];
      }
    };
    auto gather = [&](int iA, int iB, float4 (&xa)[G], float4 (&xb)[G]) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        xa[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        xb[g] = xa[g];
        if (iA >= 0) xa[g] = *reinterpret_cast<const float4*>(a.feat + (long long)iA * CIN + 16 * g + 4 * q);
        if (iB >= 0) xb[g] = *reinterpret_cast<const float4*>(a.feat + (long long)iB * CIN + 16 * g + 4 * q);
      }
    };
    // three-stage pipeline per wave: indices of offset j+2, feature rows of offset j+1, MFMAs of offset j
    int iA0, iB0, iA1, iB1, iA2, iB2;
    float4 xa[G], xb[G], ya[G], yb[G];
    fetch(0, iA0, iB0);
    fetch(1, iA1, iB1);
    gather(iA0, iB0, xa, xb);
    for (int j = 0; j < k_count; ++j) {
      fetch(j + 2, iA2, iB2);
      gather(iA1, iB1, ya, yb);
      if (__ballot(iA0 >= 0 || iB0 >= 0)) {
        const int kid = identity ? 0 : a.hdr[HDR_KOFFS + koff_begin + j];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 w = *reinterpret_cast<const float4*>(wl_s + (kid * 16 + r16) * LD + 16 * g + 4 * q);
          accA0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].x, w.x, accA0, 0, 0, 0);
          accB0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[g].x, w.x, accB0, 0, 0, 0);
          accA1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].y, w.y, accA1, 0, 0, 0);
          accB1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[g].y, w.y, accB1, 0, 0, 0);
          accA0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].z, w.z, accA0, 0, 0, 0);
          accB0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[g].z, w.z, accB0, 0, 0, 0);
          accA1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[g].w, w.w, accA1, 0, 0, 0);
          accB1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[g].w, w.w, accB1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) { xa[g] = ya[g]; xb[g] = yb[g]; }
      iA0 = iA1; iB0 = iB1; iA1 = iA2; iB1 = iB2;
    }
    // D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
    if (r16 < a.cout) {
      const float b = a.bias ? a.bias[r16] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int rA = 4 * q + e, rB = 16 + 4 * q + e;
        if (rA < npos) {
          const long long orow = a.rows ? a.rows[pos0 + rA] : pos0 + rA;
          a.out[orow * a.cout + r16] = act1(accA0[e] + accA1[e] + b, a.act, a.slope);
        }
        if (rB < npos) {
          const long long orow = a.rows ? a.rows[pos0 + rB] : pos0 + rB;
          a.out[orow * a.cout + r16] = act1(accB0[e] + accB1[e] + b, a.act, a.slope);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_conv_wave16 for 3x3x3 conv maps in canonical row order (one segment, all 27 offsets, no row list), with z-run
// reuse.  Rows are sorted with z fastest, so the dz = -1 / +1 neighbour of row r under offset (dx, dy) is, inside a
// z-run, the dz = 0 neighbour of row r -/+ 1: the lane next door already holds it.  Per (dx, dy) group a wave gathers
// the dz = 0 rows of its 16 positions once, takes the dz = -+1 operands from the adjacent lane (DPP row shift, guarded
// by index equality, so any geometry is handled) and points the loads of everything it does not need at one shared
// zero row (an L1 hit), which also removes every per-row validity branch.  PMC on the first version showed 7 VALU
// instructions per MFMA competing for the SIMD; this one is written for instruction count: 32-bit offsets, no
// identity / segment generality, tail rows clamped instead of predicated.
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
static constexpr int DPP_SHL1 = 0x101, DPP_SHR1 = 0x111;

template <int CIN>
__global__ void __launch_bounds__(512) k_conv_wave16z(Wave16Args a) {
  constexpr int LD = CIN + 4;
  constexpr int G = CIN / 16;
  constexpr int NW = 8;
  extern __shared__ __attribute__((aligned(16))) float wl_s[];   // [27][16][LD]
  for (int i = threadIdx.x; i < 27 * 16 * (CIN / 4); i += 512) {
    const int row = i / (CIN / 4), c4 = i - row * (CIN / 4);
    reinterpret_cast<float4*>(wl_s + row * LD)[c4] = reinterpret_cast<const float4*>(a.wl + (long long)row * CIN)[c4];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
  const unsigned spc = (unsigned)a.n_out;                          // one segment: positions = output rows
  const unsigned total_tiles = (spc + 15) / 16;
  const unsigned cpx = gridDim.x >> 3;
  const unsigned per_xcd = (total_tiles + 7) / 8;
  const unsigned xcd_lo = (blockIdx.x & 7) * per_xcd;
  const unsigned xcd_hi = min(total_tiles, xcd_lo + per_xcd);
  const float* wl_lane = wl_s + r16 * LD + 4 * q;

  // One (dx,dy) group of a 16-row tile: the dz=0 rows, and the dz=-+1 rows, each either the neighbouring lane's dz=0
  // row (mask k*) or loaded.  All rows come through buffer loads whose offset is out of range for an absent or
  // not-needed row: those lanes read 0 without touching memory, the number of loads in flight is fixed (exact
  // s_waitcnt distances; conditional loads made the compiler wait for the prefetch itself), and no branch is left.
  struct Grp { float4 c[G], m[G], p[G]; unsigned km, kp; };
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.feat), (short)0, (int)(unsigned)((size_t)a.n_in * CIN * 4), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;

  for (unsigned wt = xcd_lo + (blockIdx.x >> 3) * NW + (threadIdx.x >> 6); wt < xcd_hi; wt += cpx * NW) {
    const unsigned pos0 = wt * 16;
    const unsigned r = min(pos0 + r16, spc - 1);                   // tail rows repeat the last row, never stored
    const int* nb = a.nbr + r;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;

    auto issue = [&](int im, int ic, int ip, Grp& x) {
      const int cm = dpp_i<DPP_SHR1>(ic), cp = dpp_i<DPP_SHL1>(ic);
      const bool mm = im >= 0 && im == cm && r16 != 0;
      const bool mp = ip >= 0 && ip == cp && r16 != 15;
      x.km = mm ? 0xFFFFFFFFu : 0u;
      x.kp = mp ? 0xFFFFFFFFu : 0u;
      asm volatile("" : "+v"(x.km), "+v"(x.kp));       // opaque: keeps (shifted & k) | loaded as one v_and_or_b32
      const unsigned oc = ic >= 0 ? (unsigned)ic * (CIN * 4) + 16 * q : OOB;
      const unsigned om = (im >= 0 && !mm) ? (unsigned)im * (CIN * 4) + 16 * q : OOB;
      const unsigned op = (ip >= 0 && !mp) ? (unsigned)ip * (CIN * 4) + 16 * q : OOB;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        x.c[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, oc + 64 * g, 0, 0));
        x.m[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, om + 64 * g, 0, 0));
        x.p[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, op + 64 * g, 0, 0));
      }
    };
    auto mfma4 = [&](const float4& x, int slot, int g) {
      const float4 w = *reinterpret_cast<const float4*>(wl_lane + (slot * 16) * LD + 16 * g);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, w.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, w.y, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, w.z, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, w.w, acc3, 0, 0, 0);
    };
    auto mix = [](unsigned k, float shifted, float loaded) {   // (shifted & k) | loaded: loaded is 0 wherever k is set
      return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, shifted) & k) | __builtin_bit_cast(unsigned, loaded));
    };

    // pipeline per wave: indices of group g9+2, feature rows of group g9+1, MFMAs of group g9
    int im1 = nb[spc], ic1 = nb[10ull * spc], ip1 = nb[19ull * spc];                 // group 1
    Grp x, y;
    issue(nb[0], nb[9ull * spc], nb[18ull * spc], x);                                // group 0
    auto compute = [&](const Grp& x, int g9) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float4 vm, vp;
        vm.x = mix(x.km, dpp_f<DPP_SHR1>(x.c[g].x), x.m[g].x);  vp.x = mix(x.kp, dpp_f<DPP_SHL1>(x.c[g].x), x.p[g].x);
        vm.y = mix(x.km, dpp_f<DPP_SHR1>(x.c[g].y), x.m[g].y);  vp.y = mix(x.kp, dpp_f<DPP_SHL1>(x.c[g].y), x.p[g].y);
        vm.z = mix(x.km, dpp_f<DPP_SHR1>(x.c[g].z), x.m[g].z);  vp.z = mix(x.kp, dpp_f<DPP_SHL1>(x.c[g].z), x.p[g].z);
        vm.w = mix(x.km, dpp_f<DPP_SHR1>(x.c[g].w), x.m[g].w);  vp.w = mix(x.kp, dpp_f<DPP_SHL1>(x.c[g].w), x.p[g].w);
        mfma4(vm, g9, g);
        mfma4(x.c[g], g9 + 9, g);
        mfma4(vp, g9 + 18, g);
      }
    };
    // two groups per trip, the buffers swapping roles, so that no register copy ties this group's MFMAs to the
    // loads just issued for the next one (a copy made the compiler wait for them: no overlap at all)
    // (sched_barrier: the machine scheduler otherwise sinks the prefetch loads next to their first use)
#pragma unroll 1
    for (int g9 = 0; g9 < 8; g9 += 2) {
      const unsigned ga = (unsigned)(g9 + 2), gb = (unsigned)min(g9 + 3, 8);
      const int am = nb[(size_t)ga * spc], ac = nb[(size_t)(ga + 9) * spc], ap = nb[(size_t)(ga + 18) * spc];
      issue(im1, ic1, ip1, y);                                                       // group g9+1
      __builtin_amdgcn_sched_barrier(0);
      compute(x, g9);
      __builtin_amdgcn_sched_barrier(0);
      const int bm = nb[(size_t)gb * spc], bc = nb[(size_t)(gb + 9) * spc], bp = nb[(size_t)(gb + 18) * spc];
      issue(am, ac, ap, x);                                                          // group g9+2
      __builtin_amdgcn_sched_barrier(0);
      compute(y, g9 + 1);
      __builtin_amdgcn_sched_barrier(0);
      im1 = bm; ic1 = bc; ip1 = bp;
    }
    compute(x, 8);
    // D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
    if (r16 < a.cout) {
      const float b = a.bias ? a.bias[r16] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned row = pos0 + 4 * q + e;
        if (row < spc) a.out[(size_t)row * a.cout + r16] = act1(acc0[e] + acc1[e] + acc2[e] + acc3[e] + b, a.act, a.slope);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// dispatch, packing
// ------------------------------------------------------------------------------------------
static bool mfma_ok(int cin, int cout) {
  if (cout <= 4) return false;
  if (cin == 4 || cin == 8 || cin == 16) return true;
  return cin >= 32 && cin % 32 == 0;
}
static int cb_log2_for(int cin) { return cin >= 32 ? 5 : (cin == 16 ? 4 : (cin == 8 ? 3 : 2)); }
static int cout_pad_for(int cout) { const int bn = bn_for(cout); return (cout + bn - 1) / bn * bn; }

enum { KIND_NONE = -1, KIND_MFMA = 0, KIND_WAVE16 = 1, KIND_THIN_T = 2, KIND_THIN = 3 };
static int conv_kind(int K, int cin, int cout) {
  if (cout <= 4) {
    const bool pow2 = cin == 4 || cin == 8 || cin == 16 || cin == 32 || cin == 64;
    if (pow2 && (int64_t)K * cout * cin * 4 <= 48 * 1024) return KIND_THIN_T;
    return KIND_THIN;
  }
  if (cout <= 16 && (cin == 16 || cin == 32 || cin == 64) && (int64_t)K * 16 * (cin + 4) * 4 <= 64 * 1024)
    return KIND_WAVE16;
  return mfma_ok(cin, cout) ? KIND_MFMA : KIND_NONE;
}

extern "C" int64_t pcc_conv_packed_elems(int32_t K, int32_t cin, int32_t cout) {
  if (K <= 0 || cin <= 0 || cout <= 0) return 0;
  switch (conv_kind(K, cin, cout)) {
    case KIND_MFMA: return (int64_t)K * cin * cout_pad_for(cout);
    case KIND_WAVE16: return (int64_t)K * 16 * cin;
    case KIND_THIN_T: case KIND_THIN: return (int64_t)K * cin * cout;
    default: return 0;
  }
}

// scratch floats pcc_conv_fwd needs (two-pass thin form: the projection buffer t[K*cout][n_in])
extern "C" size_t pcc_conv_ws_bytes(int64_t n_in, int32_t K, int32_t cin, int32_t cout) {
  if (conv_kind(K, cin, cout) == KIND_THIN_T) return (size_t)K * cout * (size_t)n_in * sizeof(float) + 256;
  return 256;
}

// W [K][cin][cout] -> MFMA layout [K*ppo][cout_pad][CB] (zero padded columns)
__global__ void k_pack_mfma(const float* __restrict__ W, int K, int cin, int cout, int cout_pad, int cb_log2,
                            float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)K * cin * cout_pad;
  if (t >= total) return;
  const int CB = 1 << cb_log2;
  const int within = (int)(t & (CB - 1));
  const long long q = t >> cb_log2;
  const int col = (int)(q % cout_pad);
  const long long piece = q / cout_pad;
  const int ppo = cin >> cb_log2;
  const int kid = (int)(piece / ppo), cbi = (int)(piece % ppo);
  const int ci = (cbi << cb_log2) + within;
  out[t] = (col < cout) ? W[((long long)kid * cin + ci) * cout + col] : 0.f;
}

// W [K][cin][cout] -> wave16 layout [K][16][cin] (column-major per offset, zero padded to 16 columns)
__global__ void k_pack_wave16(const float* __restrict__ W, int K, int cin, int cout, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)K * 16 * cin) return;
  const int c = (int)(t % cin);
  const int o = (int)((t / cin) % 16);
  const int k = (int)(t / ((long long)cin * 16));
  out[t] = o < cout ? W[((long long)k * cin + c) * cout + o] : 0.f;
}

// W [K][cin][cout] -> thin layout [K][cout][cin]
__global__ void k_pack_thin(const float* __restrict__ W, int K, int cin, int cout, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)K * cin * cout) return;
  const int c = (int)(t % cin);
  const int o = (int)((t / cin) % cout);
  const int k = (int)(t / ((long long)cin * cout));
  out[t] = W[((long long)k * cin + c) * cout + o];
}

extern "C" int pcc_conv_pack_weights(const float* W, int32_t K, int32_t cin, int32_t cout, float* packed,
                                     int64_t packed_cap, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(W && packed && K >= 1 && K <= MAXK && cin >= 1 && cout >= 1, "pcc_conv_pack_weights: bad arguments");
  const int64_t total = pcc_conv_packed_elems(K, cin, cout);
  if (packed_cap < total) {   // a buffer sized with another layout's query (round 1: GDN sized by the conv query) is refused
    pcc_set_error("pcc_conv_pack_weights: packed buffer holds %lld floats, the layout needs %lld", (long long)packed_cap, (long long)total);
    return PCC_EWS;
  }
  const unsigned g = (unsigned)pcc_cdiv(total > 0 ? total : 1, 256);
  switch (conv_kind(K, cin, cout)) {
    case KIND_MFMA: k_pack_mfma<<<g, 256, 0, s>>>(W, K, cin, cout, cout_pad_for(cout), cb_log2_for(cin), packed); break;
    case KIND_WAVE16: k_pack_wave16<<<g, 256, 0, s>>>(W, K, cin, cout, packed); break;
    case KIND_THIN_T: case KIND_THIN: k_pack_thin<<<g, 256, 0, s>>>(W, K, cin, cout, packed); break;
    default:
      pcc_set_error("pcc_conv: unsupported shape cin=%d cout=%d (MFMA path needs cin in {4,8,16} or a multiple of 32)", cin, cout);
      return PCC_EINVAL;
  }
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ---- per-launch event timing (bench.py roofline) ------------------------------------------------
static bool g_prof_on = false;
static std::vector<hipEvent_t> g_ev_pool;
static size_t g_ev_used = 0;
static int64_t g_launches = 0;

extern "C" int pcc_prof_enable(int32_t on) {
  g_prof_on = on != 0;
  g_ev_used = 0;
  g_launches = 0;
  return PCC_OK;
}

static int prof_event(hipEvent_t* ev, hipStream_t s) {
  if (g_ev_used == g_ev_pool.size()) {
    hipEvent_t e;
    PCC_CHECK_HIP(hipEventCreate(&e));
    g_ev_pool.push_back(e);
  }
  *ev = g_ev_pool[g_ev_used++];
  PCC_CHECK_HIP(hipEventRecord(*ev, s));
  return PCC_OK;
}

extern "C" int pcc_prof_collect(double* h_conv_ms, int64_t* h_conv_launches) {
  double ms = 0.0;
  for (size_t i = 0; i + 1 < g_ev_used; i += 2) {
    PCC_CHECK_HIP(hipEventSynchronize(g_ev_pool[i + 1]));
    float t = 0.f;
    PCC_CHECK_HIP(hipEventElapsedTime(&t, g_ev_pool[i], g_ev_pool[i + 1]));
    ms += t;
  }
  if (h_conv_ms) *h_conv_ms = ms;
  if (h_conv_launches) *h_conv_launches = g_launches;
  g_ev_used = 0;
  g_launches = 0;
  return PCC_OK;
}

static bool g_mfma_buf = getenv("PCC_MFMA_BUF") ? atoi(getenv("PCC_MFMA_BUF")) != 0 : true;

template <int MODE>
static int launch_mfma(const ConvArgs& a, int tiles_bound_extra, hipStream_t s) {
  const int bn = bn_for(a.cout);
  const long long gy = a.cout_pad / bn;
  auto tiles = [&](int bm) { return (long long)(pcc_cdiv(a.n_out, bm) + tiles_bound_extra); };
  auto grid = [&](int bm) { return dim3((unsigned)((tiles(bm) * gy + 7) / 8 * 8)); };   // 1-D, multiple of 8 (XCD ranges)
  // few rows: shrink the row tile until the grid covers the 256 CUs about twice
  const long long want = 512;
  const bool buf = g_mfma_buf && a.n_in > 0 && a.n_in * a.cin * 4 <= BUF_MAX_BYTES && a.wp_elems > 0 &&
                   a.wp_elems * 4 <= BUF_MAX_BYTES;
#define PCC_LAUNCH_MFMA(WM, WN, TM, TN, BMV)                                                     \
  do {                                                                                           \
    if (buf) k_conv_mfma<WM, WN, TM, TN, MODE, true><<<grid(BMV), 256, 0, s>>>(a);               \
    else k_conv_mfma<WM, WN, TM, TN, MODE, false><<<grid(BMV), 256, 0, s>>>(a);                  \
  } while (0)
  if (bn == 128) {
    if (tiles(128) * gy >= want) PCC_LAUNCH_MFMA(2, 2, 2, 2, 128);
    else if (tiles(64) * gy >= want) PCC_LAUNCH_MFMA(2, 2, 1, 2, 64);
    else PCC_LAUNCH_MFMA(1, 4, 1, 1, 32);
  } else if (bn == 64) {
    if (tiles(128) * gy >= want) PCC_LAUNCH_MFMA(2, 2, 2, 1, 128);
    else PCC_LAUNCH_MFMA(2, 2, 1, 1, 64);
  } else PCC_LAUNCH_MFMA(4, 1, 1, 1, 128);
#undef PCC_LAUNCH_MFMA
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

static bool g_wave16_zrun = getenv("PCC_WAVE16_ZRUN") ? atoi(getenv("PCC_WAVE16_ZRUN")) != 0 : true;

template <int CIN>
static int launch_wave16(const Wave16Args& a, hipStream_t s) {
  const size_t lds = (size_t)a.K * 16 * (CIN + 4) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    PCC_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv_wave16<CIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_set = true;
  }
  const long long tiles = pcc_cdiv(a.n_out, 32) + (a.rows ? PCC_MAP_MAX_SEG : 0);
  long long want = pcc_cdiv(tiles, 8);
  want = (want + 7) / 8 * 8;                                     // multiple of 8: one contiguous tile range per XCD
  const unsigned grid = (unsigned)(want < 512 ? want : 512);     // persistent: 2 workgroups (16 waves) per CU re-use the LDS weights
  // 3x3x3 conv map in canonical row order (k_map_conv: one segment, all 27 offsets, no row list): z-run reuse variant
  if (g_wave16_zrun && a.K == 27 && a.hdr && !a.rows && a.n_out * 27 < (1ll << 31) &&
      a.n_in * CIN * 4 <= 0xFFFFFE00ll) {                          // 32-bit buffer offsets
    static bool attr_z = false;
    if (!attr_z) {
      PCC_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv_wave16z<CIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      attr_z = true;
    }
    k_conv_wave16z<CIN><<<grid, 512, lds, s>>>(a);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
  }
  k_conv_wave16<CIN><<<grid, 512, lds, s>>>(a);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

template <int CIN>
static int launch_project(const float* feat, int64_t n_in, const float* wt, int kc, float* t, hipStream_t s) {
  k_thin_project<CIN><<<(unsigned)pcc_cdiv(n_in, 256), 256, (size_t)kc * CIN * sizeof(float), s>>>(feat, n_in, wt, kc, t);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_conv_fwd(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w,
                            const float* bias, int32_t K, int32_t cout, const int32_t* hdr, const int32_t* nbr,
                            const int32_t* rows, int64_t n_out, float* out, int32_t act, float slope,
                            void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_out <= 0) return PCC_OK;
  PCC_REQUIRE(feat_in && packed_w && out && n_in > 0, "pcc_conv_fwd: NULL array");
  PCC_REQUIRE(K >= 1 && K <= MAXK, "pcc_conv_fwd: K=%d unsupported", K);
  PCC_REQUIRE(hdr ? (nbr != nullptr) : (K == 1 && n_in == n_out), "pcc_conv_fwd: map missing (only K=1 may omit it)");
  PCC_REQUIRE(act >= 0 && act <= 2, "pcc_conv_fwd: bad activation");
  PCC_REQUIRE(n_in < (1ll << 31) && n_out < (1ll << 31), "pcc_conv_fwd: too many rows");
  const int kind = conv_kind(K, cin, cout);
  PCC_REQUIRE(kind != KIND_NONE, "pcc_conv_fwd: unsupported shape cin=%d cout=%d", cin, cout);
  hipEvent_t e0, e1;
  const bool timed = g_prof_on && (kind == KIND_MFMA || kind == KIND_WAVE16);   // the roofline kernels: MFMA launches
  if (timed) PCC_TRY(prof_event(&e0, s));
  if (kind == KIND_MFMA) {
    ConvArgs a;
    a.feat = feat_in; a.wp = packed_w; a.bias = bias; a.hdr = hdr; a.nbr = nbr; a.rows = rows; a.out = out;
    a.n_out = n_out; a.cin = cin; a.cout = cout; a.cout_pad = cout_pad_for(cout);
    a.n_in = n_in; a.wp_elems = (long long)K * cin * a.cout_pad;
    a.cb_log2 = cb_log2_for(cin); a.ppo = cin >> a.cb_log2; a.act = act; a.slope = slope;
    PCC_TRY(launch_mfma<MODE_CONV>(a, rows ? PCC_MAP_MAX_SEG : 0, s));
  } else if (kind == KIND_WAVE16) {
    Wave16Args a;
    a.feat = feat_in; a.wl = packed_w; a.bias = bias; a.hdr = hdr; a.nbr = nbr; a.rows = rows; a.out = out;
    a.n_out = n_out; a.n_in = n_in; a.K = K; a.cout = cout; a.act = act; a.slope = slope;
    if (cin == 16) PCC_TRY(launch_wave16<16>(a, s));
    else if (cin == 32) PCC_TRY(launch_wave16<32>(a, s));
    else PCC_TRY(launch_wave16<64>(a, s));
  } else if (kind == KIND_THIN_T) {
    if (!ws || ws_bytes < pcc_conv_ws_bytes(n_in, K, cin, cout)) {
      pcc_set_error("pcc_conv_fwd: workspace too small (need pcc_conv_ws_bytes)");
      return PCC_EWS;
    }
    float* t = (float*)ws;
    const int kc = K * cout;
    switch (cin) {
      case 4: PCC_TRY(launch_project<4>(feat_in, n_in, packed_w, kc, t, s)); break;
      case 8: PCC_TRY(launch_project<8>(feat_in, n_in, packed_w, kc, t, s)); break;
      case 16: PCC_TRY(launch_project<16>(feat_in, n_in, packed_w, kc, t, s)); break;
      case 32: PCC_TRY(launch_project<32>(feat_in, n_in, packed_w, kc, t, s)); break;
      default: PCC_TRY(launch_project<64>(feat_in, n_in, packed_w, kc, t, s)); break;
    }
    ThinGatherArgs g;
    g.t = t; g.bias = bias; g.hdr = hdr; g.nbr = nbr; g.rows = rows; g.out = out; g.n_in = n_in; g.n_out = n_out;
    g.cout = cout; g.act = act; g.slope = slope;
    k_thin_gather<4><<<(unsigned)pcc_cdiv(n_out, 256), 256, 0, s>>>(g);
    PCC_LAUNCH_CHECK();
  } else {
    ThinArgs t;
    t.feat = feat_in; t.wt = packed_w; t.bias = bias; t.hdr = hdr; t.nbr = nbr; t.rows = rows; t.out = out;
    t.n_out = n_out; t.cin = cin; t.cout = cout; t.act = act; t.slope = slope;
    const int vec = (cin % 4 == 0) ? 4 : 1;
    int l = 0;
    while ((1 << l) < cin / vec && l < 6) ++l;
    t.lpr_log2 = l;
    const int64_t rpw = 64 >> l;
    const int64_t waves = pcc_cdiv(n_out, rpw);
    if (vec == 4) k_conv_thin<4, 4><<<(unsigned)pcc_cdiv(waves, 4), 256, 0, s>>>(t);
    else k_conv_thin<4, 1><<<(unsigned)pcc_cdiv(waves, 4), 256, 0, s>>>(t);
    PCC_LAUNCH_CHECK();
  }
  if (timed) {
    PCC_TRY(prof_event(&e1, s));
    ++g_launches;
  }
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// Pair-list form of a sparse convolution, for maps where most (offset, output row) slots are empty (5x5x5 kernels on
// surfaces: 36 of 125).  The output-stationary kernel multiplies whole 128-row tiles per active offset, so its MFMA
// work scales with K * rows, not with the pairs.  Here the pairs of each offset are compacted (padded to whole
// 128-pair tiles), T[p] = feat[in(p)] @ W[k(p)] runs as a gathered GEMM with one offset per tile -- every MFMA row is
// a real pair -- and out[o] = bias + sum_k T[pos(k, o)] is taken in ascending k: deterministic, no atomics.
// ------------------------------------------------------------------------------------------
static constexpr int PAIR_BM = 128;

__global__ void k_pair_flags(const int* __restrict__ nbr, long long n, int* __restrict__ f) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) f[e] = nbr[e] >= 0 ? 1 : 0;
}

// one block: padded start of every offset's pair range; info = {padded pairs, tiles, pairs}
__global__ void __launch_bounds__(128) k_pair_starts(const int* __restrict__ g, const int* __restrict__ nbr, long long n_out, int K,
                                                     int* __restrict__ pstart /*[K+1]*/, long long* __restrict__ info) {
  __shared__ long long cnt[MAXK];
  const long long n = n_out * K;
  const long long total = (long long)g[n - 1] + (nbr[n - 1] >= 0 ? 1 : 0);
  for (int k = threadIdx.x; k < K; k += blockDim.x) {         // the 2K boundary reads in parallel, then a short serial prefix
    const long long b = g[(long long)k * n_out];
    const long long e = (k + 1 < K) ? g[(long long)(k + 1) * n_out] : total;
    cnt[k] = e - b;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  long long run = 0;
  for (int k = 0; k < K; ++k) {
    pstart[k] = (int)run;
    run += (cnt[k] + PAIR_BM - 1) / PAIR_BM * PAIR_BM;
  }
  pstart[K] = (int)run;
  info[0] = run; info[1] = run / PAIR_BM; info[2] = total;
}

__global__ void k_pair_pos(const int* __restrict__ nbr, const int* __restrict__ g, const int* __restrict__ pstart,
                           long long n_out, int K, int* __restrict__ pos) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (o >= n_out) return;
  const long long e = (long long)k * n_out + o;
  pos[e] = nbr[e] >= 0 ? pstart[k] + (g[e] - g[(long long)k * n_out]) : -1;
}

__global__ void k_pair_fill(const int* __restrict__ nbr, const int* __restrict__ pos, long long n,
                            int* __restrict__ pair_in) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n && pos[e] >= 0) pair_in[pos[e]] = nbr[e];
}

__global__ void k_pair_tile_k(const int* __restrict__ pstart, int K, long long tiles, int* __restrict__ tile_k) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= tiles) return;
  const long long p = t * PAIR_BM;
  int lo = 0, hi = K;                 // last k with pstart[k] <= p
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pstart[mid] <= p) lo = mid; else hi = mid; }
  tile_k[t] = lo;
}

extern "C" int pcc_conv_pairs_supported(int32_t K, int32_t cin, int32_t cout) {
  return (K >= 1 && K <= MAXK && conv_kind(K, cin, cout) == KIND_MFMA && cout % 4 == 0) ? 1 : 0;
}

extern "C" size_t pcc_pair_plan_ws_bytes(int64_t n_out, int32_t K) {
  const int64_t n = n_out * K;
  return 2 * pcc_align_up((size_t)n * 4) + pcc_scan_ws_bytes(n) + 1024;
}

// phase 1: pos[k][o] (row of pair (k,o) in the padded pair list, -1 = no pair), pstart[K+1], info[3]
extern "C" int pcc_pair_plan_rank(const int32_t* nbr, int64_t n_out, int32_t K, int32_t* pos, int32_t* pstart,
                                  int64_t* info, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(nbr && pos && pstart && info && ws && n_out > 0 && K >= 1 && K <= MAXK, "pcc_pair_plan_rank: bad arguments");
  const int64_t n = n_out * K;
  PCC_REQUIRE(n + (int64_t)K * PAIR_BM < (1ll << 31), "pcc_pair_plan_rank: too many map slots");
  if (ws_bytes < pcc_pair_plan_ws_bytes(n_out, K)) { pcc_set_error("pcc_pair_plan_rank: workspace too small"); return PCC_EWS; }
  char* p = (char*)ws;
  int* f = (int*)p;  p += pcc_align_up((size_t)n * 4);
  int* g = (int*)p;  p += pcc_align_up((size_t)n * 4);
  k_pair_flags<<<(unsigned)pcc_cdiv(n, 256), 256, 0, s>>>(nbr, n, f);
  PCC_LAUNCH_CHECK();
  PCC_TRY(pcc_scan_exclusive_i32(f, g, n, p, ws_bytes - (size_t)(p - (char*)ws), s));
  k_pair_starts<<<1, 128, 0, s>>>(g, nbr, n_out, K, pstart, (long long*)info);
  PCC_LAUNCH_CHECK();
  k_pair_pos<<<dim3((unsigned)pcc_cdiv(n_out, 256), (unsigned)K), 256, 0, s>>>(nbr, g, pstart, n_out, K, pos);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// phase 2 (after the host has read info and sized the arrays): pair_in[padded pairs], tile_k[tiles]
extern "C" int pcc_pair_plan_fill(const int32_t* nbr, const int32_t* pos, const int32_t* pstart, int64_t n_out, int32_t K,
                                  int64_t padded_pairs, int32_t* pair_in, int32_t* tile_k, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(nbr && pos && pstart && pair_in && tile_k && padded_pairs % PAIR_BM == 0, "pcc_pair_plan_fill: bad arguments");
  if (padded_pairs == 0) return PCC_OK;
  PCC_CHECK_HIP(hipMemsetAsync(pair_in, 0xFF, (size_t)padded_pairs * 4, s));
  const int64_t n = n_out * K;
  k_pair_fill<<<(unsigned)pcc_cdiv(n, 256), 256, 0, s>>>(nbr, pos, n, pair_in);
  PCC_LAUNCH_CHECK();
  const int64_t tiles = padded_pairs / PAIR_BM;
  k_pair_tile_k<<<(unsigned)pcc_cdiv(tiles, 256), 256, 0, s>>>(pstart, K, tiles, tile_k);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

struct PairReduceArgs {
  const float* T; const float* bias; const int* pos; float* out; long long n_out; int K, cout, act; float slope; int lpr_log2;
};

// LPR lanes per output row, 4 channels per lane and pass; pair rows of JB offsets loaded independently
__global__ void __launch_bounds__(256) k_pair_reduce(PairReduceArgs a) {
  constexpr int JB = 5;
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << a.lpr_log2;
  const int rpw = 64 >> a.lpr_log2;
  const long long o = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + (lane >> a.lpr_log2);
  const int cl = lane & (lpr - 1);
  if (o >= a.n_out) return;
  const int cvec = a.cout / 4;
  for (int cv = cl; cv < cvec; cv += lpr) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < a.K; k0 += JB) {
      int pr[JB];
#pragma unroll
      for (int u = 0; u < JB; ++u) pr[u] = (k0 + u < a.K) ? a.pos[(long long)(k0 + u) * a.n_out + o] : -1;
      float4 x[JB];
#pragma unroll
      for (int u = 0; u < JB; ++u) {
        x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pr[u] >= 0) x[u] = reinterpret_cast<const float4*>(a.T + (long long)pr[u] * a.cout)[cv];
      }
#pragma unroll
      for (int u = 0; u < JB; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
    if (a.bias) {
      const float4 b = reinterpret_cast<const float4*>(a.bias)[cv];
      acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    }
    acc.x = act1(acc.x, a.act, a.slope); acc.y = act1(acc.y, a.act, a.slope);
    acc.z = act1(acc.z, a.act, a.slope); acc.w = act1(acc.w, a.act, a.slope);
    reinterpret_cast<float4*>(a.out + o * a.cout)[cv] = acc;
  }
}

extern "C" int pcc_conv_fwd_pairs(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w,
                                  const float* bias, int32_t K, int32_t cout, const int32_t* pair_in,
                                  const int32_t* tile_k, const int64_t* d_info, int64_t padded_pairs,
                                  const int32_t* pos, int64_t n_out, float* T, float* out, int32_t act, float slope,
                                  void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_out <= 0) return PCC_OK;
  PCC_REQUIRE(feat_in && packed_w && pair_in && tile_k && d_info && pos && T && out, "pcc_conv_fwd_pairs: NULL array");
  PCC_REQUIRE(conv_kind(K, cin, cout) == KIND_MFMA && cout % 4 == 0, "pcc_conv_fwd_pairs: shape cin=%d cout=%d not on the MFMA path", cin, cout);
  PCC_REQUIRE(padded_pairs % PAIR_BM == 0 && padded_pairs < (1ll << 31), "pcc_conv_fwd_pairs: bad pair count");
  PCC_REQUIRE(act >= 0 && act <= 2, "pcc_conv_fwd_pairs: bad activation");
  if (padded_pairs > 0) {
    ConvArgs a;
    a.feat = feat_in; a.wp = packed_w; a.bias = nullptr; a.hdr = nullptr; a.nbr = nullptr; a.rows = nullptr; a.out = T;
    a.n_out = padded_pairs; a.cin = cin; a.cout = cout; a.cout_pad = cout_pad_for(cout);
    a.n_in = n_in; a.wp_elems = (long long)K * cin * a.cout_pad;
    a.cb_log2 = cb_log2_for(cin); a.ppo = cin >> a.cb_log2; a.act = 0; a.slope = 0.f;
    a.pair_in = pair_in; a.tile_k = tile_k; a.n_tiles = (const long long*)d_info + 1;
    hipEvent_t e0, e1;
    if (g_prof_on) PCC_TRY(prof_event(&e0, s));
    const int bn = bn_for(cout);
    const long long gy = a.cout_pad / bn;
    const dim3 grid((unsigned)((padded_pairs / PAIR_BM * gy + 7) / 8 * 8));
    const bool buf = g_mfma_buf && n_in * cin * 4 <= BUF_MAX_BYTES && a.wp_elems * 4 <= BUF_MAX_BYTES;
    if (bn == 128) { if (buf) k_conv_mfma<2, 2, 2, 2, MODE_CONV, true><<<grid, 256, 0, s>>>(a); else k_conv_mfma<2, 2, 2, 2, MODE_CONV, false><<<grid, 256, 0, s>>>(a); }
    else if (bn == 64) { if (buf) k_conv_mfma<2, 2, 2, 1, MODE_CONV, true><<<grid, 256, 0, s>>>(a); else k_conv_mfma<2, 2, 2, 1, MODE_CONV, false><<<grid, 256, 0, s>>>(a); }
    else { if (buf) k_conv_mfma<4, 1, 1, 1, MODE_CONV, true><<<grid, 256, 0, s>>>(a); else k_conv_mfma<4, 1, 1, 1, MODE_CONV, false><<<grid, 256, 0, s>>>(a); }
    PCC_LAUNCH_CHECK();
    if (g_prof_on) {
      PCC_TRY(prof_event(&e1, s));
      ++g_launches;
    }
  }
  PairReduceArgs r;
  r.T = T; r.bias = bias; r.pos = pos; r.out = out; r.n_out = n_out; r.K = K; r.cout = cout; r.act = act; r.slope = slope;
  int l = 0;
  while ((1 << l) < cout / 4 && l < 6) ++l;
  r.lpr_log2 = l;
  const int64_t waves = pcc_cdiv(n_out, 64 >> l);
  k_pair_reduce<<<(unsigned)pcc_cdiv(waves, 4), 256, 0, s>>>(r);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// Transposed convolution on a SUBSET of its output rows (the rows that survive the top-k pruning), straight from their
// CSR pair lists: the P pairs are bucketed by kernel offset (LDS counting sort; the position inside a bucket does not
// matter, every T row depends on its own pair only), T[p] = feat[in(p)] @ W[k(p)] runs as the gathered pair GEMM, and
// out[o] = act(bias + sum over the row's CSR entries of T[slot(entry)]) is summed in CSR order.  Work ~ P, where the
// dense input-stationary form computes all n_in*K products and the slot-map form touches K*n_out slots.
// ------------------------------------------------------------------------------------------
static constexpr int CK_T = 256, CK_I = 8, CK_B = CK_T * CK_I;

__global__ void __launch_bounds__(CK_T) k_csr_khist(const int* __restrict__ pair_ids, const int* __restrict__ d_P, int K,
                                                    int nb, int* __restrict__ hist) {
  __shared__ int h[MAXK];
  for (int i = threadIdx.x; i < K; i += CK_T) h[i] = 0;
  __syncthreads();
  const int P = *d_P;
  const long long base = (long long)blockIdx.x * CK_B;
#pragma unroll
  for (int r = 0; r < CK_I; ++r) {
    const long long t = base + r * CK_T + threadIdx.x;
    if (t < P) atomicAdd(&h[pair_ids[t] % K], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K; i += CK_T) hist[(long long)i * nb + blockIdx.x] = h[i];
}

__global__ void __launch_bounds__(128) k_csr_kstarts(const int* __restrict__ off, const int* __restrict__ d_P, int K, int nb,
                                                     int* __restrict__ pstart, long long* __restrict__ info) {
  __shared__ long long cnt[MAXK];
  const long long total = *d_P;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const long long b = off[(long long)k * nb];
    const long long e = (k + 1 < K) ? off[(long long)(k + 1) * nb] : total;
    cnt[k] = e - b;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  long long run = 0;
  for (int k = 0; k < K; ++k) {
    pstart[k] = (int)run;
    run += (cnt[k] + PAIR_BM - 1) / PAIR_BM * PAIR_BM;
  }
  pstart[K] = (int)run;
  info[0] = run; info[1] = run / PAIR_BM; info[2] = total;
}

__global__ void __launch_bounds__(CK_T) k_csr_kscatter(const int* __restrict__ pair_ids, const int* __restrict__ d_P, int K,
                                                       int nb, const int* __restrict__ off, const int* __restrict__ pstart,
                                                       int* __restrict__ pair_in, int* __restrict__ slot) {
  __shared__ int cur[MAXK];
  for (int i = threadIdx.x; i < K; i += CK_T)
    cur[i] = pstart[i] + off[(long long)i * nb + blockIdx.x] - off[(long long)i * nb];
  __syncthreads();
  const int P = *d_P;
  const long long base = (long long)blockIdx.x * CK_B;
#pragma unroll
  for (int r = 0; r < CK_I; ++r) {
    const long long t = base + r * CK_T + threadIdx.x;
    if (t < P) {
      const int pid = pair_ids[t];
      const int i = pid / K, k = pid - i * K;
      const int pos = atomicAdd(&cur[k], 1);
      pair_in[pos] = i;
      slot[t] = pos;
    }
  }
}

struct CsrReduceArgs {
  const float* T; const float* bias; const int* first; const int* slot; float* out; long long n_out;
  int cout, act; float slope; int lpr_log2;
};

__global__ void __launch_bounds__(256) k_csr_reduce(CsrReduceArgs a) {
  constexpr int JB = 4;
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << a.lpr_log2;
  const int rpw = 64 >> a.lpr_log2;
  const long long o = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + (lane >> a.lpr_log2);
  const int cl = lane & (lpr - 1);
  if (o >= a.n_out) return;
  const int cvec = a.cout / 4;
  const int t0 = a.first[o], t1 = a.first[o + 1];
  for (int cv = cl; cv < cvec; cv += lpr) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = t0; t < t1; t += JB) {
      int sl[JB];
#pragma unroll
      for (int u = 0; u < JB; ++u) sl[u] = (t + u < t1) ? a.slot[t + u] : -1;
      float4 x[JB];
#pragma unroll
      for (int u = 0; u < JB; ++u) {
        x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sl[u] >= 0) x[u] = reinterpret_cast<const float4*>(a.T + (long long)sl[u] * a.cout)[cv];
      }
#pragma unroll
      for (int u = 0; u < JB; ++u) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
    }
    if (a.bias) {
      const float4 b = reinterpret_cast<const float4*>(a.bias)[cv];
      acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    }
    acc.x = act1(acc.x, a.act, a.slope); acc.y = act1(acc.y, a.act, a.slope);
    acc.z = act1(acc.z, a.act, a.slope); acc.w = act1(acc.w, a.act, a.slope);
    reinterpret_cast<float4*>(a.out + o * a.cout)[cv] = acc;
  }
}

// pairs: host value of first[n_out] (the number of CSR entries).  Scratch: int_ws and T sized by the two queries.
extern "C" size_t pcc_convt_rows_int_ws_bytes(int64_t pairs, int32_t K) {
  const int64_t nb = pcc_cdiv(pairs > 0 ? pairs : 1, CK_B);
  const int64_t padded = pairs + (int64_t)K * PAIR_BM;
  return pcc_align_up((size_t)K * nb * 4) + pcc_align_up((size_t)padded * 4) + pcc_align_up((size_t)(pairs + 1) * 4) +
         pcc_align_up((size_t)(padded / PAIR_BM + 1) * 4) + pcc_align_up((size_t)(K + 1) * 4) + 64 +
         pcc_scan_ws_bytes((int64_t)K * nb) + 1024;
}
extern "C" int64_t pcc_convt_rows_t_elems(int64_t pairs, int32_t K, int32_t cout) {
  return (pairs + (int64_t)K * PAIR_BM) * cout;
}

extern "C" int pcc_convt_fwd_rows(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w,
                                  const float* bias, int32_t K, int32_t cout, const int32_t* first,
                                  const int32_t* pair_ids, int64_t n_out, int64_t pairs, float* T, float* out,
                                  int32_t act, float slope, void* int_ws, size_t int_ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_out <= 0) return PCC_OK;
  PCC_REQUIRE(feat_in && packed_w && first && pair_ids && T && out && int_ws, "pcc_convt_fwd_rows: NULL array");
  PCC_REQUIRE(K >= 1 && K <= MAXK && conv_kind(K, cin, cout) == KIND_MFMA && cout % 4 == 0,
              "pcc_convt_fwd_rows: shape K=%d cin=%d cout=%d not on the MFMA path", K, cin, cout);
  PCC_REQUIRE(pairs >= 0 && pairs + (int64_t)K * PAIR_BM < (1ll << 31) && act >= 0 && act <= 2, "pcc_convt_fwd_rows: bad arguments");
  if (int_ws_bytes < pcc_convt_rows_int_ws_bytes(pairs, K)) { pcc_set_error("pcc_convt_fwd_rows: workspace too small"); return PCC_EWS; }
  const int64_t nb = pcc_cdiv(pairs > 0 ? pairs : 1, CK_B);
  const int64_t padded_cap = pairs + (int64_t)K * PAIR_BM;
  char* p = (char*)int_ws;
  int* hist = (int*)p;        p += pcc_align_up((size_t)K * nb * 4);
  int* pair_in = (int*)p;     p += pcc_align_up((size_t)padded_cap * 4);
  int* slot = (int*)p;        p += pcc_align_up((size_t)(pairs + 1) * 4);
  int* tile_k = (int*)p;      p += pcc_align_up((size_t)(padded_cap / PAIR_BM + 1) * 4);
  int* pstart = (int*)p;      p += pcc_align_up((size_t)(K + 1) * 4);
  long long* info = (long long*)p;  p += 64;
  void* scan_ws = p;
  const size_t scan_bytes = int_ws_bytes - (size_t)(p - (char*)int_ws);
  const int* d_P = first + n_out;
  k_csr_khist<<<(unsigned)nb, CK_T, 0, s>>>(pair_ids, d_P, K, (int)nb, hist);
  PCC_LAUNCH_CHECK();
  PCC_TRY(pcc_scan_exclusive_i32(hist, hist, (int64_t)K * nb, scan_ws, scan_bytes, s));
  k_csr_kstarts<<<1, 128, 0, s>>>(hist, d_P, K, (int)nb, pstart, info);
  PCC_LAUNCH_CHECK();
  PCC_CHECK_HIP(hipMemsetAsync(pair_in, 0xFF, (size_t)padded_cap * 4, s));
  k_csr_kscatter<<<(unsigned)nb, CK_T, 0, s>>>(pair_ids, d_P, K, (int)nb, hist, pstart, pair_in, slot);
  PCC_LAUNCH_CHECK();
  const int64_t tiles_cap = padded_cap / PAIR_BM;
  k_pair_tile_k<<<(unsigned)pcc_cdiv(tiles_cap, 256), 256, 0, s>>>(pstart, K, tiles_cap, tile_k);
  PCC_LAUNCH_CHECK();
  {
    ConvArgs a;
    a.feat = feat_in; a.wp = packed_w; a.bias = nullptr; a.hdr = nullptr; a.nbr = nullptr; a.rows = nullptr; a.out = T;
    a.n_out = padded_cap; a.cin = cin; a.cout = cout; a.cout_pad = cout_pad_for(cout);
    a.n_in = n_in; a.wp_elems = (long long)K * cin * a.cout_pad;
    a.cb_log2 = cb_log2_for(cin); a.ppo = cin >> a.cb_log2; a.act = 0; a.slope = 0.f;
    a.pair_in = pair_in; a.tile_k = tile_k; a.n_tiles = info + 1;
    hipEvent_t e0, e1;
    if (g_prof_on) PCC_TRY(prof_event(&e0, s));
    const int bn = bn_for(cout);
    const long long gy = a.cout_pad / bn;
    const dim3 grid((unsigned)((tiles_cap * gy + 7) / 8 * 8));
    const bool buf = g_mfma_buf && n_in * cin * 4 <= BUF_MAX_BYTES && a.wp_elems * 4 <= BUF_MAX_BYTES;
    if (bn == 128) { if (buf) k_conv_mfma<2, 2, 2, 2, MODE_CONV, true><<<grid, 256, 0, s>>>(a); else k_conv_mfma<2, 2, 2, 2, MODE_CONV, false><<<grid, 256, 0, s>>>(a); }
    else if (bn == 64) { if (buf) k_conv_mfma<2, 2, 2, 1, MODE_CONV, true><<<grid, 256, 0, s>>>(a); else k_conv_mfma<2, 2, 2, 1, MODE_CONV, false><<<grid, 256, 0, s>>>(a); }
    else { if (buf) k_conv_mfma<4, 1, 1, 1, MODE_CONV, true><<<grid, 256, 0, s>>>(a); else k_conv_mfma<4, 1, 1, 1, MODE_CONV, false><<<grid, 256, 0, s>>>(a); }
    PCC_LAUNCH_CHECK();
    if (g_prof_on) {
      PCC_TRY(prof_event(&e1, s));
      ++g_launches;
    }
  }
  CsrReduceArgs r;
  r.T = T; r.bias = bias; r.first = first; r.slot = slot; r.out = out; r.n_out = n_out; r.cout = cout; r.act = act; r.slope = slope;
  int l = 0;
  while ((1 << l) < cout / 4 && l < 6) ++l;
  r.lpr_log2 = l;
  const int64_t waves = pcc_cdiv(n_out, 64 >> l);
  k_csr_reduce<<<(unsigned)pcc_cdiv(waves, 4), 256, 0, s>>>(r);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// Generative transposed convolution, input stationary.
//   Every (input row i, kernel offset k) is exactly one pair of the map, so the products
//   T[i][k][:] = feat[i] @ W[k] form ONE dense GEMM  [n_in, cin] x [cin, K*cout]  with no gather and no padding
//   waste, however sparse the output neighbourhoods are.  The sum over the pairs of an output row is then taken
//   in fixed order (class offsets ascending) through the transposed map: deterministic, no atomics.
// ------------------------------------------------------------------------------------------
__global__ void k_pack_convt(const float* __restrict__ W, int K, int cin, int cout, int ncol, int cout_pad,
                             int cb_log2, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)cin * cout_pad;
  if (t >= total) return;
  const int CB = 1 << cb_log2;
  const int within = (int)(t & (CB - 1));
  const long long q = t >> cb_log2;
  const int col = (int)(q % cout_pad);
  const int cbi = (int)(q / cout_pad);
  const int ci = (cbi << cb_log2) + within;
  float v = 0.f;
  if (col < ncol) {
    const int k = col / cout, co = col - k * cout;
    v = W[((long long)k * cin + ci) * cout + co];
  }
  out[t] = v;
}

extern "C" int64_t pcc_convt_packed_elems(int32_t K, int32_t cin, int32_t cout) {
  if (K <= 0 || cin <= 0 || cout <= 0 || !mfma_ok(cin, K * cout)) return 0;
  return (int64_t)cin * cout_pad_for(K * cout);
}

extern "C" int pcc_convt_pack_weights(const float* W, int32_t K, int32_t cin, int32_t cout, float* packed,
                                      int64_t packed_cap, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(W && packed && K >= 1 && K <= MAXK_T && cin >= 1 && cout >= 1, "pcc_convt_pack_weights: bad arguments");
  PCC_REQUIRE(mfma_ok(cin, K * cout), "pcc_convt: unsupported shape cin=%d (needs 4, 8, 16 or a multiple of 32)", cin);
  const int64_t total = pcc_convt_packed_elems(K, cin, cout);
  if (packed_cap < total) {
    pcc_set_error("pcc_convt_pack_weights: packed buffer holds %lld floats, the layout needs %lld", (long long)packed_cap, (long long)total);
    return PCC_EWS;
  }
  k_pack_convt<<<(unsigned)pcc_cdiv(total, 256), 256, 0, s>>>(W, K, cin, cout, K * cout, cout_pad_for(K * cout),
                                                             cb_log2_for(cin), packed);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

struct GatherArgs {
  const float* T; const float* bias; const int* hdr; const int* nbr; const int* rows;
  float* out; long long n_out; int K, cout, act; float slope; int lpr_log2;
};

// LPR lanes per output position, VEC channels per lane and pass; offsets in batches of independent loads
template <int VEC>
__global__ void __launch_bounds__(256) k_convt_gather(GatherArgs a) {
  typedef typename ThinVec<VEC>::T VT;
  constexpr int JB = 9;
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << a.lpr_log2;
  const int rpw = 64 >> a.lpr_log2;
  const long long p = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + (lane >> a.lpr_log2);
  const int cl = lane & (lpr - 1);
  if (p >= a.n_out) return;
  const int cvec = a.cout / VEC;
  const int nseg = a.hdr[HDR_NSEG];
  int s = 0;
  for (; s < nseg - 1; ++s) {
    const int* sg = a.hdr + HDR_SEG0 + s * SEG_WORDS;
    if (p < (long long)sg[SEG_POS_BEGIN] + sg[SEG_POS_COUNT]) break;
  }
  const int* sg = a.hdr + HDR_SEG0 + s * SEG_WORDS;
  const int k_count = sg[SEG_K_COUNT], koff_begin = sg[SEG_KOFF_BEGIN];
  const long long spc = sg[SEG_POS_COUNT], local = p - sg[SEG_POS_BEGIN];
  const int* seg_nbr = a.nbr + (((long long)(unsigned)sg[SEG_NBR_LO]) | ((long long)sg[SEG_NBR_HI] << 32));
  const long long orow = a.rows ? a.rows[p] : p;
  for (int cv = cl; cv < cvec; cv += lpr) {
    VT acc;
    thin_zero(acc);
    for (int j0 = 0; j0 < k_count; j0 += JB) {
      int ir[JB];
#pragma unroll
      for (int u = 0; u < JB; ++u) ir[u] = (j0 + u < k_count) ? seg_nbr[(long long)(j0 + u) * spc + local] : -1;
      VT x[JB];
#pragma unroll
      for (int u = 0; u < JB; ++u) {
        thin_zero(x[u]);
        if (ir[u] >= 0) {
          const int kid = a.hdr[HDR_KOFFS + koff_begin + j0 + u];
          x[u] = reinterpret_cast<const VT*>(a.T + ((long long)ir[u] * a.K + kid) * a.cout)[cv];
        }
      }
#pragma unroll
      for (int u = 0; u < JB; ++u) thin_acc(acc, x[u]);     // fixed order: offsets ascending
    }
    VT b;
    thin_zero(b);
    if (a.bias) b = reinterpret_cast<const VT*>(a.bias)[cv];
    thin_acc(acc, b);
    thin_act(acc, a.act, a.slope);
    reinterpret_cast<VT*>(a.out + orow * a.cout)[cv] = acc;
  }
}

extern "C" int pcc_convt_fwd(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w, const float* bias,
                             int32_t K, int32_t cout, const int32_t* hdr, const int32_t* nbr, const int32_t* rows,
                             int64_t n_out, float* T, float* out, int32_t act, float slope, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_out <= 0 || n_in <= 0) return PCC_OK;
  PCC_REQUIRE(feat_in && packed_w && hdr && nbr && rows && T && out, "pcc_convt_fwd: NULL array");
  PCC_REQUIRE(K >= 1 && K <= MAXK && mfma_ok(cin, K * cout), "pcc_convt_fwd: unsupported shape K=%d cin=%d cout=%d", K, cin, cout);
  PCC_REQUIRE(act >= 0 && act <= 2, "pcc_convt_fwd: bad activation");
  PCC_REQUIRE(n_in < (1ll << 31) && n_out < (1ll << 31), "pcc_convt_fwd: too many rows");
  // 1) dense GEMM  T[n_in, K*cout] = feat[n_in, cin] @ Wflat[cin, K*cout]
  ConvArgs a;
  a.feat = feat_in; a.wp = packed_w; a.bias = nullptr; a.hdr = nullptr; a.nbr = nullptr; a.rows = nullptr; a.out = T;
  a.n_out = n_in; a.cin = cin; a.cout = K * cout; a.cout_pad = cout_pad_for(K * cout);
  a.n_in = n_in; a.wp_elems = (long long)cin * a.cout_pad;
  a.cb_log2 = cb_log2_for(cin); a.ppo = cin >> a.cb_log2; a.act = 0; a.slope = 0.f;
  hipEvent_t e0, e1;
  if (g_prof_on) PCC_TRY(prof_event(&e0, s));
  PCC_TRY(launch_mfma<MODE_CONV>(a, 0, s));
  if (g_prof_on) {
    PCC_TRY(prof_event(&e1, s));
    ++g_launches;
  }
  // 2) ordered gather-sum through the transposed map
  GatherArgs g;
  g.T = T; g.bias = bias; g.hdr = hdr; g.nbr = nbr; g.rows = rows; g.out = out; g.n_out = n_out; g.K = K; g.cout = cout;
  g.act = act; g.slope = slope;
  const int vec = (cout % 4 == 0) ? 4 : 1;
  int l = 0;
  while ((1 << l) < cout / vec && l < 6) ++l;
  g.lpr_log2 = l;
  const int64_t waves = pcc_cdiv(n_out, 64 >> l);
  if (vec == 4) k_convt_gather<4><<<(unsigned)pcc_cdiv(waves, 4), 256, 0, s>>>(g);
  else k_convt_gather<1><<<(unsigned)pcc_cdiv(waves, 4), 256, 0, s>>>(g);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// CSR form of the generative transposed convolution: out[o] = act(bias + sum_{t in [first[o], first[o+1])} T[pair_ids[t]])
// (pair lists from pcc_coords_expand_csr; outputs are written in canonical row order, no `rows` indirection).
struct GatherCsrArgs {
  const float* T; const float* bias; const int* first; const int* pair_ids;
  float* out; long long n_out; int cout, act; float slope; int lpr_log2;
  const int* ex_nbr; const float* ex_bias; int ex_K;      // optional: + sum over the existing neighbours k of ex_bias[k]
};

template <int VEC>
__global__ void __launch_bounds__(256) k_convt_gather_csr(GatherCsrArgs a) {
  typedef typename ThinVec<VEC>::T VT;
  constexpr int JB = 8;
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << a.lpr_log2;
  const int rpw = 64 >> a.lpr_log2;
  const long long o = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + (lane >> a.lpr_log2);
  const int cl = lane & (lpr - 1);
  if (o >= a.n_out) return;
  const int cvec = a.cout / VEC;
  const int t0 = a.first[o], t1 = a.first[o + 1];
  for (int cv = cl; cv < cvec; cv += lpr) {
    VT acc;
    thin_zero(acc);
    for (int t = t0; t < t1; t += JB) {
      int pid[JB];
#pragma unroll
      for (int u = 0; u < JB; ++u) pid[u] = (t + u < t1) ? a.pair_ids[t + u] : -1;
      VT x[JB];
#pragma unroll
      for (int u = 0; u < JB; ++u) {
        thin_zero(x[u]);
        if (pid[u] >= 0) x[u] = reinterpret_cast<const VT*>(a.T + (long long)pid[u] * a.cout)[cv];
      }
#pragma unroll
      for (int u = 0; u < JB; ++u) thin_acc(acc, x[u]);     // fixed order: pair id ascending
    }
    if (a.ex_nbr) {              // neighbour k of this row exists -> its constant contribution (fused affine layers)
      for (int k = 0; k < a.ex_K; ++k)
        if (a.ex_nbr[(long long)k * a.n_out + o] >= 0) thin_acc(acc, reinterpret_cast<const VT*>(a.ex_bias + (long long)k * a.cout)[cv]);
    }
    VT b;
    thin_zero(b);
    if (a.bias) b = reinterpret_cast<const VT*>(a.bias)[cv];
    thin_acc(acc, b);
    thin_act(acc, a.act, a.slope);
    reinterpret_cast<VT*>(a.out + o * a.cout)[cv] = acc;
  }
}

extern "C" int pcc_convt_fwd_csr(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w,
                                 const float* bias, int32_t K, int32_t cout, const int32_t* first,
                                 const int32_t* pair_ids, int64_t n_out, float* T, float* out, int32_t act, float slope,
                                 const int32_t* ex_nbr, int32_t ex_K, const float* ex_bias, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_out <= 0 || n_in <= 0) return PCC_OK;
  PCC_REQUIRE(feat_in && packed_w && first && pair_ids && T && out, "pcc_convt_fwd_csr: NULL array");
  PCC_REQUIRE(K >= 1 && K <= MAXK_T && mfma_ok(cin, K * cout), "pcc_convt_fwd_csr: unsupported shape K=%d cin=%d cout=%d", K, cin, cout);
  PCC_REQUIRE(!ex_nbr || (ex_bias && ex_K >= 1), "pcc_convt_fwd_csr: ex_nbr needs ex_bias and ex_K");
  PCC_REQUIRE(act >= 0 && act <= 2, "pcc_convt_fwd_csr: bad activation");
  PCC_REQUIRE(n_in * K < (1ll << 31) && n_out < (1ll << 31), "pcc_convt_fwd_csr: too many rows");
  ConvArgs a;
  a.feat = feat_in; a.wp = packed_w; a.bias = nullptr; a.hdr = nullptr; a.nbr = nullptr; a.rows = nullptr; a.out = T;
  a.n_out = n_in; a.cin = cin; a.cout = K * cout; a.cout_pad = cout_pad_for(K * cout);
  a.n_in = n_in; a.wp_elems = (long long)cin * a.cout_pad;
  a.cb_log2 = cb_log2_for(cin); a.ppo = cin >> a.cb_log2; a.act = 0; a.slope = 0.f;
  hipEvent_t e0, e1;
  if (g_prof_on) PCC_TRY(prof_event(&e0, s));
  PCC_TRY(launch_mfma<MODE_CONV>(a, 0, s));
  if (g_prof_on) {
    PCC_TRY(prof_event(&e1, s));
    ++g_launches;
  }
  GatherCsrArgs g;
  g.T = T; g.bias = bias; g.first = first; g.pair_ids = pair_ids; g.out = out; g.n_out = n_out; g.cout = cout;
  g.act = act; g.slope = slope; g.ex_nbr = ex_nbr; g.ex_bias = ex_bias; g.ex_K = ex_K;
  const int vec = (cout % 4 == 0) ? 4 : 1;
  int l = 0;
  while ((1 << l) < cout / vec && l < 6) ++l;
  g.lpr_log2 = l;
  const int64_t waves = pcc_cdiv(n_out, 64 >> l);
  if (vec == 4) k_convt_gather_csr<4><<<(unsigned)pcc_cdiv(waves, 4), 256, 0, s>>>(g);
  else k_convt_gather_csr<1><<<(unsigned)pcc_cdiv(waves, 4), 256, 0, s>>>(g);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// GDN
// ------------------------------------------------------------------------------------------
__global__ void k_gdn_pack(const float* __restrict__ beta_raw, const float* __restrict__ gamma_raw, int c,
                           float beta_bound, float gamma_bound, float pedestal, int cout_pad, int cb_log2,
                           float* __restrict__ packed, float* __restrict__ beta_eff) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < c) {
    const float b = fmaxf(beta_raw[t], beta_bound);
    beta_eff[t] = b * b - pedestal;
  }
  const long long total = (long long)c * cout_pad;
  if (t >= total) return;
  // conv weight W[ci][co] = gamma[co][ci]; packed layout [ppo][cout_pad][CB]
  const int CB = 1 << cb_log2;
  const int within = (int)(t & (CB - 1));
  const long long q = t >> cb_log2;
  const int col = (int)(q % cout_pad);
  const int cbi = (int)(q / cout_pad);
  const int ci = (cbi << cb_log2) + within;
  float v = 0.f;
  if (col < c) {
    const float g = fmaxf(gamma_raw[(long long)col * c + ci], gamma_bound);
    v = g * g - pedestal;
  }
  packed[t] = v;
}

extern "C" int64_t pcc_gdn_packed_elems(int32_t c) { return mfma_ok(c, c) ? (int64_t)c * cout_pad_for(c) : 0; }

extern "C" int pcc_gdn_pack(const float* beta_raw, const float* gamma_raw, int32_t c, float beta_min, float* packed,
                            int64_t packed_cap, float* beta_eff, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(beta_raw && gamma_raw && packed && beta_eff, "pcc_gdn_pack: NULL array");
  PCC_REQUIRE(mfma_ok(c, c), "pcc_gdn: channel count %d unsupported (needs 8, 16 or a multiple of 32)", c);
  const double pedestal = 1.0 / 68719476736.0;   // 2^-36
  const float beta_bound = (float)sqrt((double)beta_min + pedestal);
  const float gamma_bound = (float)sqrt(pedestal);
  const int64_t total = pcc_gdn_packed_elems(c);
  if (packed_cap < total) {
    pcc_set_error("pcc_gdn_pack: packed buffer holds %lld floats, the layout needs %lld", (long long)packed_cap, (long long)total);
    return PCC_EWS;
  }
  k_gdn_pack<<<(unsigned)pcc_cdiv(total, 256), 256, 0, s>>>(beta_raw, gamma_raw, c, beta_bound, gamma_bound,
                                                           (float)pedestal, cout_pad_for(c), cb_log2_for(c), packed,
                                                           beta_eff);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_gdn_fwd(const float* x, int64_t n, int32_t c, const float* packed, const float* beta_eff,
                           int32_t inverse, float* out, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(x && packed && beta_eff && out && x != out, "pcc_gdn_fwd: bad arguments");
  PCC_REQUIRE(mfma_ok(c, c), "pcc_gdn: channel count %d unsupported", c);
  ConvArgs a;
  a.feat = x; a.wp = packed; a.bias = beta_eff; a.hdr = nullptr; a.nbr = nullptr; a.rows = nullptr; a.out = out;
  a.n_out = n; a.cin = c; a.cout = c; a.cout_pad = cout_pad_for(c);
  a.n_in = n; a.wp_elems = (long long)c * a.cout_pad;
  a.cb_log2 = cb_log2_for(c); a.ppo = c >> a.cb_log2; a.act = 0; a.slope = 0.f;
  if (inverse) return launch_mfma<MODE_IGDN>(a, 0, s);
  return launch_mfma<MODE_GDN>(a, 0, s);
}
