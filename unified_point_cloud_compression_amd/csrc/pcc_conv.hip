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
// Phase switches of the GEMM kernels (tools/gemm_probe.py): compiled in only by `make DBG=1` (-DPCC_DBG_BUILD); in the shipped
// library the tests are the constant 0 and the compiler drops them.
#ifdef PCC_DBG_BUILD
#define PCC_DBG_ON(a, bit) (((a).dbg & (bit)) != 0)
#else
#define PCC_DBG_ON(a, bit) false
#endif
static constexpr int PAIR_BM_C = 128;   // rows per pair tile (pair-list GEMMs)
__device__ inline float act1(float v, int act, float slope);

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
  const unsigned char* featb = nullptr; // split path: bf16 planes of feat, [n_in][cin/32][3][32] (k_feat_split)
  int ksplit = 1;                       // split path, map mode: the (offset, channel-block) reduction cut over ksplit workgroups
  float* part = nullptr;                //   partial tiles [ksplit][n_out][cout], summed in fixed order by k_splitk_reduce
  int dbg = 0;                          // diagnostics (probe builds only, `make DBG=1` + env PCC_DBG): 1 = no output stores, 2 = no MFMA phase, 4 = no staging loads
  int nt = 0;                           // non-temporal accesses of streamed buffers (g_nt): 1 = dense products' stores, 2 = pair products' stores
  bool wh_ok = false;                   // dense products: the pack carries scaled fp16 planes + column scales (split_planes_h)
  const unsigned char* feath = nullptr; //   scaled fp16 planes of feat, [n_in][cin/32][2][32] (k_feat_split_h)
  const float* frow_inv = nullptr;      //   and 1 / (power-of-two scale) of every feature row
  int arith = PCC_ARITH_H3;             // arithmetic form of this call (include/pcc_hip.h PCC_ARITH_*): an argument of every entry point, no process state
  int* guard = nullptr;                 //   range guard of the fp16-pair products (the entry point's d_guard): set to 1 when a (row, column) pair of
  float guard_lim = 0.f;                //   a tile has rinv * cinv * 8 * cin > guard_lim, i.e. max|row| * max|column| * cin * 2^-27 may exceed the budget
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
  int tile_id = wid / gy;
  int colblock = (wid - tile_id * gy) * BN;
  if (a.hdr == nullptr && a.pair_in == nullptr && gy > 8) {
    // dense GEMM with many column blocks (generative transposed convs: [n_in, cin] x [cin, K*cout], weights > L2):
    // groups of 8 row tiles sweep the column blocks together, so a block's weights are fetched once per group instead of
    // once per row tile (the grid covers whole groups, launch_mfma)
    const int g = wid / (8 * gy), rem = wid - g * 8 * gy;
    colblock = (rem >> 3) * BN;
    tile_id = g * 8 + (rem & 7);
  }

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
// The same implicit GEMM on the bf16 matrix pipe at fp32 accuracy ("split" path, cin a multiple of 32).
//
// gfx950 runs fp32-input MFMAs at the vector rate (157 TFLOP/s) and bf16-input MFMAs 16 times faster.  Every fp32
// operand is split EXACTLY into three bf16 values, x = h + m + l (h = rne_bf16(x), m = rne_bf16(x - h),
// l = x - h - m: 8 + 8 + 8 mantissa bits and a sign each, both subtractions exact), and a product is evaluated as the
// six cross terms of first and second order,
//     x*w ~= l*h' + h*l' + m*m' + m*h' + h*m' + h*h'        (dropped: m*l' + l*m' + l*l' <= 2^-23 |x*w|)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- the same fp32 accumulation the
// fp32 MFMA performs.  Six bf16 MFMAs of K = 16 replace eight fp32 MFMAs of K = 2 at a quarter of the cycles each
// (MI355X_MICROARCH.md: 32 against 64 cycles per SIMD): 2.67x the matrix throughput, error at the level of fp32
// rounding (tests/test_gpu_map_conv.py::test_split_path_accuracy compares both paths with a float64 evaluation).
// Weights are split once when packed (three bf16 planes behind the fp32 image); the feature rows of a convolution's
// input are split by one pass of k_feat_split into planes [row][cin/32][3][32] (library scratch), so the MFMA kernel
// stages pure 16-byte copies.  (First version: split while staging, 5.5 VALU operations per element -- SQ counters
// showed 5.4 VALU instructions per MFMA and the SIMD issue-bound at 33 % MFMA utilisation; and for the shallow
// generative GEMMs every column block repeated the split of the same rows.)
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__host__ __device__ inline long long bf_plane_elems(long long fp32_elems) { return fp32_elems / 2 * 3; }   // floats holding 3 bf16 planes

__device__ __forceinline__ void bf_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const f32x2v v = {x0, x1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
  const f32x2v r1 = {x0 - __builtin_bit_cast(float, h << 16), x1 - __builtin_bit_cast(float, h & 0xFFFF0000u)};
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2v));
  const f32x2v r2 = {r1.x - __builtin_bit_cast(float, m << 16), r1.y - __builtin_bit_cast(float, m & 0xFFFF0000u)};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2v));
}

// fp32 packed image [rows][32] -> bf16 planes [rows][3][32]
__global__ void k_split_packed(const float* __restrict__ src, long long pairs, unsigned* __restrict__ dst) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one pair of consecutive channels
  if (t >= pairs) return;
  const long long row = t >> 4;
  const int cp = (int)(t & 15);
  unsigned h, m, l;
  bf_split2(src[2 * t], src[2 * t + 1], h, m, l);
  unsigned* d = dst + row * 48 + cp;
  d[0] = h; d[16] = m; d[32] = l;
}

// feature rows fp32 [n][c] -> bf16 planes [n][c/32][3][32] (of |x| for the GDN modes: the split is odd, so the planes of
// -x are the negated planes of x).  One pass per convolution input, so that the MFMA kernel stages pure copies.
__global__ void k_feat_split(const float* __restrict__ x, long long pairs, int cpairs, int take_abs, unsigned* __restrict__ dst) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one pair of consecutive channels
  if (t >= pairs) return;
  const long long row = t / cpairs;
  const int cp = (int)(t - row * cpairs);
  float2 v = reinterpret_cast<const float2*>(x)[t];
  if (take_abs) { v.x = fabsf(v.x); v.y = fabsf(v.y); }
  unsigned h, m, l;
  bf_split2(v.x, v.y, h, m, l);
  unsigned* d = dst + (row * (cpairs >> 4) + (cp >> 4)) * 48 + (cp & 15);
  d[0] = h; d[16] = m; d[32] = l;
}

// ---- scaled fp16 pairs (dense products of the generative transposed convolutions) ----------------------------------
// T = X W with every fp32 operand written as s^-1 (h + l): s a power of two that brings the row's (X) or column's (W)
// largest magnitude into [2^14, 2^15), h = fp16(s x), l = fp16(s x - h).  h + l carries >= 22 significant bits of every
// element within 2^-18 of its row / column maximum (smaller ones lose bits they could not contribute to an fp32 sum anyway),
// and the three products l*h, h*l, h*h (fp32 accumulate, v_mfma_f32_32x32x16_f16) leave out only l*l < 2^-22 of a term: the
// error stays at the level of the fp32 accumulation itself (tests/test_gpu_map_conv.py::test_dense_products_accuracy), at
// HALF the matrix instructions of the six-term bf16 form and 2/3 of its operand bytes.  Row scales factor out of a dense
// product (one input row per output row) but not out of a gathered convolution, which is why only the dense products take
// this form.  The kernels are bound by energy, not by issue slots: the chip holds ~1.3 GHz on them (GRBM_GUI_ACTIVE / 8 /
// wall), and every phase's cost adds up whether or not it overlaps, so fewer MFMAs is what pays.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float pow2_scale_exp(float mx, int& e) {      // s = 2^(14 - e), e = floor(log2 mx) (0 for mx = 0)
  e = mx > 0.f ? ilogbf(mx) : 0;
  e = e < -100 ? -100 : (e > 120 ? 120 : e);
  return ldexpf(1.f, 14 - e);
}
__device__ __forceinline__ void h_split4(const float4 v, float s, f16x4& h, f16x4& l) {
  const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = (_Float16)x[i];
    l[i] = (_Float16)(x[i] - (float)h[i]);
  }
}

// fp32 packed GEMM image [piece][cout_pad][32] -> fp16 planes [piece][cout_pad][2][32] scaled per column, and 1/scale per column
// (blockIdx.y: kernel offset of a convolution pack [K*ppo][cout_pad][32]; 0 for the flat operand of a dense product)
__global__ void k_split_packed_h(const float* __restrict__ src, int ppo, int cout_pad, unsigned char* __restrict__ dst,
                                 float* __restrict__ cinv) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= cout_pad) return;
  src += (size_t)blockIdx.y * ppo * cout_pad * 32;
  dst += (size_t)blockIdx.y * ppo * cout_pad * 128;
  cinv += (size_t)blockIdx.y * cout_pad;
  float mx = 0.f;
  for (int pc = 0; pc < ppo; ++pc) {
    const float4* r = reinterpret_cast<const float4*>(src + ((size_t)pc * cout_pad + col) * 32);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = r[q];
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
  }
  int e;
  const float sc = pow2_scale_exp(mx, e);
  for (int pc = 0; pc < ppo; ++pc) {
    const float4* r = reinterpret_cast<const float4*>(src + ((size_t)pc * cout_pad + col) * 32);
    unsigned char* d = dst + ((size_t)pc * cout_pad + col) * 128;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      f16x4 h, l;
      h_split4(r[q], sc, h, l);
      *reinterpret_cast<f16x4*>(d + q * 8) = h;
      *reinterpret_cast<f16x4*>(d + 64 + q * 8) = l;
    }
  }
  cinv[col] = ldexpf(1.f, e - 14);
}

// feature rows fp32 [n][c] -> fp16 planes [n][c/32][2][32] scaled per row, and 1/scale per row.  G = 2^g_log2 lanes per row.
__global__ void __launch_bounds__(256) k_feat_split_h(const float* __restrict__ x, long long n, int cin, int g_log2,
                                                      unsigned char* __restrict__ dst, float* __restrict__ rinv) {
  const int lane = threadIdx.x & 63;
  const int G = 1 << g_log2, lg = lane & (G - 1);
  const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 >> g_log2) + (lane >> g_log2);
  const bool live = row < n;
  const int c4 = cin >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x) + (live ? row : 0) * c4;
  float4 v[2];
  float mx = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int q = lg + it * G;
    v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && q < c4) v[it] = xr[q];
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[it].x), fabsf(v[it].y)), fmaxf(fabsf(v[it].z), fabsf(v[it].w))));
  }
  for (int d = G >> 1; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
  int e;
  const float sc = pow2_scale_exp(mx, e);
  if (!live) return;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int q = lg + it * G;
    if (q >= c4) continue;
    f16x4 h, l;
    h_split4(v[it], sc, h, l);
    unsigned char* d = dst + ((size_t)row * (cin >> 5) + (q >> 3)) * 128 + (q & 7) * 8;
    *reinterpret_cast<f16x4*>(d) = h;
    *reinterpret_cast<f16x4*>(d + 64) = l;
  }
  if (lg == 0) rinv[row] = ldexpf(1.f, e - 14);
}

// Grow-only device scratch of the library (per device; stream-ordered reuse on the caller's stream): the bf16 planes of
// the current convolution's input.  The only memory libpcc_hip owns.
static void* g_scratch[64];
static size_t g_scratch_bytes[64];
static int lib_scratch(size_t bytes, void** out) {
  int dev = 0;
  PCC_CHECK_HIP(hipGetDevice(&dev));
  dev &= 63;
  if (g_scratch_bytes[dev] < bytes) {
    if (g_scratch[dev]) { PCC_CHECK_HIP(hipDeviceSynchronize()); PCC_CHECK_HIP(hipFree(g_scratch[dev])); g_scratch[dev] = nullptr; g_scratch_bytes[dev] = 0; }
    const size_t want = bytes + bytes / 4 + (1 << 20);
    PCC_CHECK_HIP(hipMalloc(&g_scratch[dev], want));
    g_scratch_bytes[dev] = want;
  }
  *out = g_scratch[dev];
  return PCC_OK;
}
// a second, small grow-only scratch (tables that live beside the planes of the same call)
static void* g_scratch_small[64];
static size_t g_scratch_small_bytes[64];
static int lib_scratch_small(size_t bytes, void** out) {
  int dev = 0;
  PCC_CHECK_HIP(hipGetDevice(&dev));
  dev &= 63;
  if (g_scratch_small_bytes[dev] < bytes) {
    if (g_scratch_small[dev]) { PCC_CHECK_HIP(hipDeviceSynchronize()); PCC_CHECK_HIP(hipFree(g_scratch_small[dev])); g_scratch_small[dev] = nullptr; g_scratch_small_bytes[dev] = 0; }
    const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    PCC_CHECK_HIP(hipMalloc(&g_scratch_small[dev], want));
    g_scratch_small_bytes[dev] = want;
  }
  *out = g_scratch_small[dev];
  return PCC_OK;
}
static int make_planes(ConvArgs& a, bool take_abs, hipStream_t s) {
  void* p = nullptr;
  PCC_TRY(lib_scratch((size_t)a.n_in * a.cin * 6, &p));
  const long long pairs = (long long)a.n_in * a.cin / 2;
  k_feat_split<<<(unsigned)pcc_cdiv(pairs, 256), 256, 0, s>>>(a.feat, pairs, a.cin / 2, take_abs ? 1 : 0, (unsigned*)p);
  PCC_LAUNCH_CHECK();
  a.featb = (const unsigned char*)p;
  return PCC_OK;
}

// Range guard of the fp16-pair products: a device word (the entry point's d_guard) the kernels OR 1 into when a tile's scales
// admit an absolute product error above PCC_H_GUARD_BUDGET (cin * 2^-27 * max|row| * max|column| > budget); NULL = no guard.
// The caller zeroes the word, reads it back with a size it reads anyway, and repeats the operation with PCC_ARITH_BF6 when it is
// set.  The form and the guard word are ARGUMENTS of every call: the library keeps no arithmetic state (round 4; the process-wide
// pcc_set_gemm_h / pcc_set_mfma_split / pcc_set_h_guard switches of rounds 2-3 are gone).
static int set_arith(ConvArgs& a, int arith, int32_t* d_guard, const char* who) {
  if (arith < PCC_ARITH_F32 || arith > PCC_ARITH_H3) { pcc_set_error("%s: arith=%d is not a PCC_ARITH_* form", who, arith); return PCC_EINVAL; }
  a.arith = arith;
  a.guard = arith == PCC_ARITH_H3 ? d_guard : nullptr;
  a.guard_lim = PCC_H_GUARD_BUDGET;              // compared with cin * 2^-27 * (rinv * 2^15) * (cinv * 2^15) = rinv * cinv * 8 * cin
  return PCC_OK;
}

static int make_planes_h(ConvArgs& a, hipStream_t s) {
  PCC_REQUIRE(a.cin % 32 == 0 && a.cin <= 512, "dense products: cin=%d (needs a multiple of 32 up to 512)", a.cin);
  void* p = nullptr;
  const size_t plane_bytes = pcc_align_up((size_t)a.n_in * a.cin * 4);
  PCC_TRY(lib_scratch(plane_bytes + pcc_align_up((size_t)a.n_in * 4), &p));
  int g = 0;
  while ((1 << g) < a.cin / 4 && g < 6) ++g;
  const long long rows_per_block = 4ll * (64 >> g);
  k_feat_split_h<<<(unsigned)pcc_cdiv(a.n_in, rows_per_block), 256, 0, s>>>(a.feat, a.n_in, a.cin, g, (unsigned char*)p,
                                                                         (float*)((char*)p + plane_bytes));
  PCC_LAUNCH_CHECK();
  a.feath = (const unsigned char*)p;
  a.frow_inv = (const float*)((char*)p + plane_bytes);
  return PCC_OK;
}

// (launch bounds: the 128 x 128 tile needs ~240 registers; capped at 168 for three workgroups per CU it spilled 96 bytes per
//  thread and reloaded loop-invariant offsets inside the chunk loop.  With two workgroups per CU nothing spills; measured equal
//  (282 against 287 us on the last hyper-synthesis layer: that launch is bound by its L2 operand traffic).)
template <int WM, int WN, int TM, int TN, int MODE, int MINWG = (TM * TN >= 4 ? 2 : 3)>
__global__ void __launch_bounds__(256, MINWG) k_conv_mfma_bf(ConvArgs a) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(BN <= 128 && BM <= 128, "one feature row and one weight row per thread (pair)");
  // LDS images [row][13 x 16 B]: the 12 units (plane, slot) of a row's 192-byte piece plus one unit of padding.  Staging
  // writes go 8 consecutive units at a time (contiguous), a fragment read takes one unit of 16 different rows: row * 52
  // dwords mod 64 is a permutation of the bank quads over the rows of any ds_read_b128 lane group.  Conflict-free both ways.
  constexpr int LDU = 13;
  __shared__ __attribute__((aligned(16))) uint4 As[BM * LDU];
  __shared__ __attribute__((aligned(16))) uint4 Bs[BN * LDU];
  __shared__ unsigned char act_flag[MAXK];
  __shared__ unsigned char act_list[MAXK];
  __shared__ unsigned char act_kid[MAXK];
  __shared__ int s_nact;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int cpx = gridDim.x >> 3;
  int wid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
  const int ks_id = wid % a.ksplit;                   // which slice of the reduction (ksplit == 1: the whole of it)
  wid /= a.ksplit;
  const int gy = a.cout_pad / BN;
  int tile_id = wid / gy;
  int colblock = (wid - tile_id * gy) * BN;
  if (a.hdr == nullptr && a.pair_in == nullptr && gy > 8) {
    // dense GEMM with many column blocks (generative transposed convs: [n_in, cin] x [cin, K*cout], weights > L2):
    // groups of 8 row tiles sweep the column blocks together, so a block's weights are fetched once per group instead of
    // once per row tile (the grid covers whole groups, launch_mfma)
    const int g = wid / (8 * gy), rem = wid - g * 8 * gy;
    colblock = (rem >> 3) * BN;
    tile_id = g * 8 + (rem & 7);
  }

  int pos0, npos, k_count, koff_begin;
  long long seg_pos_count;
  const int* seg_nbr = nullptr;
  const bool pair_mode = (a.pair_in != nullptr);
  const bool identity = (a.hdr == nullptr) && !pair_mode;
  if (pair_mode) {
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
    int tile = tile_id, sgi = 0;
    bool found = false;
    int pb = 0, pc = 0;
    for (; sgi < nseg; ++sgi) {
      const int* sg = a.hdr + HDR_SEG0 + sgi * SEG_WORDS;
      pb = sg[SEG_POS_BEGIN]; pc = sg[SEG_POS_COUNT];
      const int tiles = (pc + BM - 1) / BM;
      if (tile < tiles) { found = true; break; }
      tile -= tiles;
    }
    if (!found) return;
    const int* sg = a.hdr + HDR_SEG0 + sgi * SEG_WORDS;
    k_count = sg[SEG_K_COUNT];
    koff_begin = sg[SEG_KOFF_BEGIN];
    const long long nb = ((long long)(unsigned)sg[SEG_NBR_LO]) | ((long long)sg[SEG_NBR_HI] << 32);
    seg_nbr = a.nbr + nb;
    seg_pos_count = pc;
    const int local0 = tile * BM;
    pos0 = pb + local0;
    npos = min(BM, pc - local0);
    seg_nbr += local0;
  }

  if (pair_mode) {
    if (tid == 0) { act_list[0] = 0; act_kid[0] = (unsigned char)a.tile_k[tile_id]; s_nact = 1; }
  } else if (identity) {
    if (tid == 0) { act_list[0] = 0; act_kid[0] = 0; s_nact = 1; }
  } else {
    for (int j = w; j < k_count; j += 4) {
      bool any = false;
      for (int r = lane; r < npos; r += 64) any |= (seg_nbr[(long long)j * seg_pos_count + r] >= 0);
      const unsigned long long mk = __ballot(any);
      if (lane == 0) act_flag[j] = mk ? 1 : 0;
    }
    __syncthreads();
    if (w == 0) {
      int nn = 0;
      for (int j0 = 0; j0 < k_count; j0 += 64) {
        const int u = j0 + lane;
        const int j = (u < k_count) ? a.hdr[HDR_ORDER + koff_begin + u] : 0;
        const bool f = (u < k_count) && act_flag[j];
        const unsigned long long mk = __ballot(f);
        if (f) {
          const int p = nn + __popcll(mk & ((1ull << lane) - 1ull));
          act_list[p] = (unsigned char)j;
          act_kid[p] = (unsigned char)a.hdr[HDR_KOFFS + koff_begin + j];
        }
        nn += __popcll(mk);
      }
      if (lane == 0) s_nact = nn;
    }
  }
  __syncthreads();
  const int nact = s_nact;
  const int nchunks_all = nact * a.ppo;               // CB = 32: one piece per chunk
  // split-K: slice ks_id takes the chunks [c_lo, c_hi) (contiguous: whole offsets stay together as far as possible)
  const int c_lo = (int)((long long)nchunks_all * ks_id / a.ksplit), c_hi = (int)((long long)nchunks_all * (ks_id + 1) / a.ksplit);
  const int nchunks = c_hi - c_lo;

  // staging roles: the 16-byte units u = j * 256 + tid of the tile's piece, 12 per row (3 planes x 4 slots), rows contiguous:
  // consecutive lanes read consecutive 16-byte units of a feature / weight row (coalesced), and write them side by side
  constexpr int NA = (BM * 12 + 255) / 256, NB = (BN * 12 + 255) / 256;
  int a_row[NA], a_w[NA], b_row[NB], b_w[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) { const int u = j * 256 + tid; a_row[j] = u / 12; a_w[j] = u - a_row[j] * 12; if (u >= BM * 12) a_row[j] = -1; }
#pragma unroll
  for (int j = 0; j < NB; ++j) { const int u = j * 256 + tid; b_row[j] = u / 12; b_w[j] = u - b_row[j] * 12; if (u >= BN * 12) b_row[j] = -1; }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm = w / WN, wn = w % WN;
  const int half = lane >> 5, r31 = lane & 31;

  const unsigned row_bytes = (unsigned)a.cin * 6u;    // [cin/32][3][32] bf16
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(a.featb), (short)0, (int)(unsigned)((size_t)a.n_in * row_bytes), 0x00020000);
  const float* wb = a.wp + a.wp_elems;                // bf16 planes behind the fp32 image: [piece][cout_pad][3][32] bf16
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wb), (short)0, (int)(unsigned)((size_t)bf_plane_elems(a.wp_elems) * 4), 0x00020000);

  auto load_rows = [&](int ai, int (&rows)[NA]) {
    const int slot = act_list[ai];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int rc = min(max(a_row[j], 0), npos - 1);          // tail rows repeat the tile's last row (never stored)
      rows[j] = identity ? (pos0 + rc) : seg_nbr[(long long)slot * seg_pos_count + rc];
      if (a_row[j] < 0) rows[j] = -1;
    }
  };
  auto issue = [&](int ai, int cbi, const int (&rows)[NA], uint4 (&av)[NA], uint4 (&bv)[NB]) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const unsigned off = rows[j] >= 0 ? (unsigned)rows[j] * row_bytes + (unsigned)cbi * 192u + (unsigned)a_w[j] * 16u : BUF_OOB;
      av[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, PCC_DBG_ON(a, 4) ? BUF_OOB : off, 0, 0));
    }
    const unsigned wbase = (unsigned)((act_kid[ai] * a.ppo + cbi) * a.cout_pad + colblock) * 192u;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const unsigned off = b_row[j] >= 0 ? wbase + (unsigned)(j * 256 + tid) * 16u : BUF_OOB;
      bv[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, PCC_DBG_ON(a, 4) ? BUF_OOB : off, 0, 0));
    }
  };

  int rows_cur[NA], rows_nxt[NA];
  uint4 av[NA], bv[NB];
  int ai_c = 0, cbi_c = 0, ai_n = 0, cbi_n = 0;
  if (nchunks > 0) {
    ai_c = c_lo / a.ppo; cbi_c = c_lo - ai_c * a.ppo;
    load_rows(ai_c, rows_cur);
    issue(ai_c, cbi_c, rows_cur, av, bv);
    if (nchunks > 1) {
      ai_n = (c_lo + 1) / a.ppo; cbi_n = (c_lo + 1) - ai_n * a.ppo;
      if (ai_n != ai_c) load_rows(ai_n, rows_nxt);
      else {
#pragma unroll
        for (int j = 0; j < NA; ++j) rows_nxt[j] = rows_cur[j];
      }
    }
  }

  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();   // previous chunk's fragment reads are done
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if (a_row[j] >= 0) As[a_row[j] * LDU + a_w[j]] = av[j];
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (b_row[j] >= 0) Bs[b_row[j] * LDU + b_w[j]] = bv[j];
    __syncthreads();
    if (c + 1 < nchunks) {          // next chunk's global loads fly during this chunk's MFMAs
#pragma unroll
      for (int j = 0; j < NA; ++j) rows_cur[j] = rows_nxt[j];
      ai_c = ai_n; cbi_c = cbi_n;
      issue(ai_c, cbi_c, rows_cur, av, bv);
      if (c + 2 < nchunks) {
        ai_n = (c_lo + c + 2) / a.ppo; cbi_n = (c_lo + c + 2) - ai_n * a.ppo;
        if (ai_n != ai_c) load_rows(ai_n, rows_nxt);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs, not next to its use
    if (PCC_DBG_ON(a, 2)) continue;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[p][i] = __builtin_bit_cast(bf16x8, As[((wm * TM + i) * 32 + r31) * LDU + p * 4 + ks * 2 + half]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bf[p][j] = __builtin_bit_cast(bf16x8, Bs[((wn * TN + j) * 32 + r31) * LDU + p * 4 + ks * 2 + half]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {           // smallest terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
        }
    }
  }

  if (PCC_DBG_ON(a, 1)) { if (acc[0][0][0] != 12345.678f) return; }
  if (a.ksplit > 1) {                                 // raw partial sums; bias / activation are applied by k_splitk_reduce
    float* const part = a.part + (size_t)ks_id * (size_t)a.n_out * a.cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = colblock + (wn * TN + j) * 32 + r31;
      if (col >= a.cout) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
          if (r >= npos) continue;
          const long long orow = a.rows ? a.rows[pos0 + r] : (pos0 + r);
          part[orow * a.cout + col] = acc[i][j][e];
        }
    }
    return;
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
// Dense GEMM form of the split kernel, stripped to what the products of the generative transposed convolutions need:
//   T[n, ncol] = X[n, cin] x W[cin, ncol],  cin = NCH * 32, no bias / activation / row list, 128 x 128 tiles.
// Same data flow as k_conv_mfma_bf (bf16 planes -> registers -> padded LDS images -> six MFMA terms, fp32 accumulate), but a
// tile here is only NCH = 4 chunks deep, so the fixed cost per tile decided the run time of the general kernel: with loads,
// MFMAs and stores all switched off it still took 0.77 of 2.65 ms on the level-2 products (PCC_DBG) --
// tile decode through the map header, per-chunk offset arithmetic for gathered rows, 64-bit address arithmetic for each of the
// 64 stores of a lane.  Here every address is (per-tile scalar base in a buffer descriptor) + (per-lane offset computed once)
// + (compile-time immediate or a scalar), the chunk loop is unrolled, and tail tiles take their own path.
// ------------------------------------------------------------------------------------------
template <int NCH>
__global__ void __launch_bounds__(256, 3) k_gemm_bf2(ConvArgs a) {
  constexpr int BM = 128, BN = 128, LDU = 13;
  constexpr unsigned ROWB = NCH * 192u;                // bytes of a feature row's planes
  __shared__ __attribute__((aligned(16))) uint4 As[BM * LDU];
  __shared__ __attribute__((aligned(16))) uint4 Bs[BN * LDU];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int cpx = gridDim.x >> 3;
  const int wid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
  const int gy = a.cout_pad / BN;
  int tile_id, colblock;
  if (gy > 8) {                                         // groups of 8 row tiles sweep the column blocks together (weights > L2)
    const int g = wid / (8 * gy), rem = wid - g * 8 * gy;
    colblock = (rem >> 3) * BN;
    tile_id = g * 8 + (rem & 7);
  } else {
    tile_id = wid / gy;
    colblock = (wid - tile_id * gy) * BN;
  }
  const long long p0 = (long long)tile_id * BM;
  if (p0 >= a.n_out) return;
  const int npos = (int)min((long long)BM, a.n_out - p0);

  // descriptors: the tile's feature rows (rows past the end read as zero), the column block's weights, the tile's output rows
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(a.featb) + (size_t)p0 * ROWB, (short)0, (int)((unsigned)npos * ROWB), 0x00020000);
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.wp + a.wp_elems) + (size_t)colblock * 192u;
  const unsigned b_stride = (unsigned)a.cout_pad * 192u;             // bytes between the weight planes of consecutive chunks
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(wb), (short)0, (int)((NCH - 1) * b_stride + BN * 192u), 0x00020000);

  // staging roles: 16-byte unit u = j * 256 + tid of the tile's [128 rows][12 units] piece, j = 0..5
  unsigned vA[6], ld[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const unsigned u = (unsigned)(j * 256 + tid), row = u / 12u, wu = u - row * 12u;
    vA[j] = row * ROWB + wu * 16u;
    ld[j] = row * LDU + wu;
  }
  const unsigned vB = (unsigned)tid * 16u;

  uint4 av[6], bv[6];
  auto issue = [&](int cbi) {
#pragma unroll
    for (int j = 0; j < 6; ++j)
      av[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, PCC_DBG_ON(a, 4) ? BUF_OOB : vA[j] + (unsigned)cbi * 192u, 0, 0));
#pragma unroll
    for (int j = 0; j < 6; ++j)
      bv[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, PCC_DBG_ON(a, 4) ? BUF_OOB : vB, (int)((unsigned)cbi * b_stride + (unsigned)j * 4096u), 0));
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm = w >> 1, wn = w & 1;
  const int half = lane >> 5, r31 = lane & 31;
  const unsigned fa = (unsigned)((wm * 64 + r31) * LDU + half), fb = (unsigned)((wn * 64 + r31) * LDU + half);

  issue(0);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    __syncthreads();   // previous chunk's fragment reads are done
#pragma unroll
    for (int j = 0; j < 6; ++j) As[ld[j]] = av[j];
#pragma unroll
    for (int j = 0; j < 6; ++j) Bs[ld[j]] = bv[j];
    __syncthreads();
    if (c + 1 < NCH) issue(c + 1);            // next chunk's global loads fly during this chunk's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    if (PCC_DBG_ON(a, 2)) continue;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[3][2], bf[3][2];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) af[p][i] = __builtin_bit_cast(bf16x8, As[fa + i * 32 * LDU + p * 4 + ks * 2]);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[p][j] = __builtin_bit_cast(bf16x8, Bs[fb + j * 32 * LDU + p * 4 + ks * 2]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {           // smallest terms first (same order as k_conv_mfma_bf: identical results)
          if (!PCC_DBG_ON(a, 8)) {                  // (timing experiment: three of the six terms)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
        }
    }
  }
  if (PCC_DBG_ON(a, 1)) { if (acc[0][0][0] != 12345.678f) return; }

  // ---- range guard: the elements of a row / column far below its maximum are carried with an
  //      ABSOLUTE error of 2^-28 of that maximum, so a product's error can reach cin * 2^-27 * max|row| * max|column|; the
  //      scales bound the maxima (max < 2^15 / scale).  A lane's rows x a lane's columns are exactly its outputs.
  // (evaluated on the row scales the epilogue reads anyway)
  // ---- stores: element e of acc[i][j] is row wm*64 + i*32 + (e&3) + 8*(e>>2) + 4*half, column wn*64 + j*32 + r31 of the tile
  const unsigned ncol = (unsigned)a.cout;
  float* const obase = a.out + (size_t)p0 * ncol + colblock;
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
      obase, (short)0, (int)(((unsigned)(npos - 1) * ncol + min((unsigned)BN, ncol - (unsigned)colblock)) * 4u), 0x00020000);
  const unsigned vO = ((unsigned)(wm * 64 + 4 * half) * ncol + (unsigned)(wn * 64 + r31)) * 4u;
  if (npos == BM && (unsigned)colblock + BN <= ncol) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned so = (unsigned)(i * 32 + (e & 3) + 8 * (e >> 2)) * ncol * 4u;      // scalar
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float v = acc[i][j][e];          // (a bit_cast of the vector element itself compiles to element 0)
          if ((a.nt & 1) && NCH >= 2) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsO, vO + (unsigned)j * 128u, (int)so, 2);   // non-temporal, as k_gemm_h2
          else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsO, vO + (unsigned)j * 128u, (int)so, 0);
        }
      }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if ((unsigned)colblock + (unsigned)(wn * 64 + j * 32 + r31) >= ncol) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (r >= npos) continue;
        obase[(size_t)r * ncol + (unsigned)(wn * 64 + j * 32 + r31)] = acc[i][j][e];
      }
  }
}

// GDN / IGDN with the split folded into the staging (round 3): out = x / (beta + |x| gamma^T)  (or x * (...)).  The general
// kernel above reads bf16 planes that a k_feat_split pass wrote first -- for this K = C, one-row-per-row product that pass and
// the planes' round trip are more HBM traffic than the operation itself (205 k x 128 rows: 50 us split + 170 us product for
// 205 MB of algorithmic traffic).  Here a thread loads the fp32 rows, takes |x| and splits in registers (the same bf_split2,
// so planes, term order and chunk order are those of k_conv_mfma_bf: bit-identical results) and writes the LDS image itself.
// TM = 2: 128-row tiles, TM = 1: 64-row tiles (more workgroups for the mid-sized sets).
template <int TM, int MODE>
__global__ void __launch_bounds__(256, 3) k_gdn_bf(ConvArgs a) {
  static_assert(MODE == MODE_GDN || MODE == MODE_IGDN, "GDN modes only");
  constexpr int WM = 2, WN = 2, TN = 2;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, LDU = 13;
  __shared__ __attribute__((aligned(16))) uint4 As[BM * LDU];
  __shared__ __attribute__((aligned(16))) uint4 Bs[BN * LDU];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int cpx = gridDim.x >> 3;
  const int wid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
  const int gy = a.cout_pad / BN;
  const int tile_id = wid / gy;
  const int colblock = (wid - tile_id * gy) * BN;
  const long long p0 = (long long)tile_id * BM;
  if (p0 >= a.n_out) return;
  const int pos0 = (int)p0;
  const int npos = (int)min((long long)BM, a.n_out - p0);
  const int nchunks = a.ppo;                                          // 32 channels per chunk

  constexpr int NX = BM * 8 / 256, NB = (BN * 12 + 255) / 256;      // float4 of x / 16-byte weight units per thread and chunk
  int x_row[NX], x_j[NX], b_row[NB], b_w[NB];
#pragma unroll
  for (int q = 0; q < NX; ++q) { const int u = q * 256 + tid; x_row[q] = u >> 3; x_j[q] = u & 7; }
#pragma unroll
  for (int q = 0; q < NB; ++q) { const int u = q * 256 + tid; b_row[q] = u / 12; b_w[q] = u - b_row[q] * 12; if (u >= BN * 12) b_row[q] = -1; }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int wm = w / WN, wn = w % WN;
  const int half = lane >> 5, r31 = lane & 31;

  const float* wb = a.wp + a.wp_elems;                                // bf16 planes behind the fp32 image
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wb), (short)0, (int)(unsigned)((size_t)bf_plane_elems(a.wp_elems) * 4), 0x00020000);
  const float* const xt = a.feat + (size_t)pos0 * a.cin;

  float4 xv[NX];
  uint4 bv[NB];
  auto issue = [&](int cbi) {
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int rc = min(x_row[q], npos - 1);                         // tail rows repeat the tile's last row (never stored)
      xv[q] = *reinterpret_cast<const float4*>(xt + (size_t)rc * a.cin + cbi * 32 + x_j[q] * 4);
    }
    const unsigned wbase = (unsigned)(cbi * a.cout_pad + colblock) * 192u;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const unsigned off = b_row[q] >= 0 ? wbase + (unsigned)(q * 256 + tid) * 16u : BUF_OOB;
      bv[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, off, 0, 0));
    }
  };
  issue(0);
  unsigned long long* const As64 = reinterpret_cast<unsigned long long*>(As);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();   // previous chunk's fragment reads are done
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      unsigned h0, m0, l0, h1, m1, l1;
      bf_split2(fabsf(xv[q].x), fabsf(xv[q].y), h0, m0, l0);
      bf_split2(fabsf(xv[q].z), fabsf(xv[q].w), h1, m1, l1);
      const int o = x_row[q] * (LDU * 2) + x_j[q];                    // 8-byte slots: plane p of the row starts at slot 8 p
      As64[o] = (unsigned long long)h0 | ((unsigned long long)h1 << 32);
      As64[o + 8] = (unsigned long long)m0 | ((unsigned long long)m1 << 32);
      As64[o + 16] = (unsigned long long)l0 | ((unsigned long long)l1 << 32);
    }
#pragma unroll
    for (int q = 0; q < NB; ++q)
      if (b_row[q] >= 0) Bs[b_row[q] * LDU + b_w[q]] = bv[q];
    __syncthreads();
    if (c + 1 < nchunks) issue(c + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[p][i] = __builtin_bit_cast(bf16x8, As[((wm * TM + i) * 32 + r31) * LDU + p * 4 + ks * 2 + half]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bf[p][j] = __builtin_bit_cast(bf16x8, Bs[((wn * TN + j) * 32 + r31) * LDU + p * 4 + ks * 2 + half]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {           // smallest terms first (the order of k_conv_mfma_bf)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
        }
    }
  }
  // ---- epilogue: out = x / (beta + acc)  or  x * (beta + acc) --------------------------------------------------------
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = colblock + (wn * TN + j) * 32 + r31;
    if (col >= a.cout) continue;
    const float b = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (r >= npos) continue;
        const size_t o = (size_t)(pos0 + r) * a.cout + col;
        const float v = acc[i][j][e] + b;
        const float x = a.feat[o];
        a.out[o] = (MODE == MODE_GDN) ? x / v : x * v;
      }
  }
}

// The input layer (4 -> 128 channels, 5x5x5, stride 2: `g_a.down_conv_1[0]`) with its (offset, channel) pairs flattened into
// ONE reduction axis of K * 4 <= 512 (round 3).  The general kernel walks the 125 offsets one at a time -- a 4-deep reduction per
// staging round, 125 rounds of gather -> LDS -> barrier -> MFMA per tile: 2.8 us per round of pure latency, 22 TFLOP/s.  Here a
// chunk is 8 offsets x 4 channels = the 32-wide piece the split kernels work on: a thread gathers the 16-byte feature rows of
// (output row, offset) pairs through the map, splits them into bf16 planes in registers (as k_gdn_bf does) and writes the LDS
// image; 16 rounds instead of 125, six bf16 MFMA terms per product.  Absent neighbours are out-of-range buffer loads (zeros).
// Weights: planes [16 pieces][cout_pad][3][32] bf16 of the flattened kernel, converted from the packed fp32 image per call
// (k_in4_weight_planes, 0.4 MB).  Map indices are fetched two chunks ahead, rows one chunk ahead.
__global__ void __launch_bounds__(256) k_in4_weight_planes(const float* __restrict__ wp /*[K][cout_pad][4]*/, int K, int cout_pad,
                                                           unsigned* __restrict__ planes /*[16][cout_pad][48 dwords]*/) {
  const int t = blockIdx.x * 256 + threadIdx.x;                       // one pair of consecutive flat indices of one column
  if (t >= 16 * cout_pad * 16) return;
  const int cp = t & 15, col = (t >> 4) % cout_pad, piece = t / (16 * cout_pad);
  const int f0 = piece * 32 + 2 * cp;                                 // flat index = 4 * offset + channel
  const int k = f0 >> 2, c = f0 & 3;                                  // (f0 even: both elements of the pair belong to offset k)
  float v0 = 0.f, v1 = 0.f;
  if (k < K) { const float* w = wp + ((size_t)k * cout_pad + col) * 4 + c; v0 = w[0]; v1 = w[1]; }
  unsigned h, m, l;
  bf_split2(v0, v1, h, m, l);
  unsigned* d = planes + ((size_t)piece * cout_pad + col) * 48 + cp;
  d[0] = h; d[16] = m; d[32] = l;
}

template <int TM>
__global__ void __launch_bounds__(256, 3) k_conv_in4_bf(ConvArgs a, const unsigned* __restrict__ wplanes, int K) {
  constexpr int WM = 2, WN = 2, TN = 2;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, LDU = 13, NCHUNK = 16;
  __shared__ __attribute__((aligned(16))) uint4 As[BM * LDU];
  __shared__ __attribute__((aligned(16))) uint4 Bs[BN * LDU];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int cpx = gridDim.x >> 3;
  const int wid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
  const int gy = a.cout_pad / BN;
  const int tile_id = wid / gy;
  const int colblock = (wid - tile_id * gy) * BN;
  const long long p0 = (long long)tile_id * BM;
  if (p0 >= a.n_out) return;
  const int pos0 = (int)p0;
  const int npos = (int)min((long long)BM, a.n_out - p0);
  const int nchunks = (K * 4 + 31) / 32;                              // <= NCHUNK

  constexpr int NX = BM * 8 / 256, NB = (BN * 12 + 255) / 256;
  int x_row[NX], x_j[NX], b_row[NB], b_w[NB];
#pragma unroll
  for (int q = 0; q < NX; ++q) { const int u = q * 256 + tid; x_row[q] = u >> 3; x_j[q] = u & 7; }
#pragma unroll
  for (int q = 0; q < NB; ++q) { const int u = q * 256 + tid; b_row[q] = u / 12; b_w[q] = u - b_row[q] * 12; if (u >= BN * 12) b_row[q] = -1; }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int wm = w / WN, wn = w % WN;
  const int half = lane >> 5, r31 = lane & 31;

  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned*>(wplanes), (short)0, (int)(unsigned)((size_t)NCHUNK * a.cout_pad * 192u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.feat), (short)0, (int)(unsigned)((size_t)a.n_in * 16u), 0x00020000);
  const int* const nb0 = a.nbr + pos0;

  int id[NX];
  uint4 xv[NX], bv[NB];
  auto load_idx = [&](int c) {
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int k = c * 8 + x_j[q];
      id[q] = (k < K && x_row[q] < npos) ? nb0[(long long)k * a.n_out + x_row[q]] : -1;
    }
  };
  auto issue = [&](int c) {                                           // rows of chunk c (indices already in id[]) + its weights
#pragma unroll
    for (int q = 0; q < NX; ++q)
      xv[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsX, id[q] >= 0 ? (unsigned)id[q] * 16u : BUF_OOB, 0, 0));
    const unsigned wbase = (unsigned)(c * a.cout_pad + colblock) * 192u;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const unsigned off = b_row[q] >= 0 ? wbase + (unsigned)(q * 256 + tid) * 16u : BUF_OOB;
      bv[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, off, 0, 0));
    }
  };
  load_idx(0);
  issue(0);
  if (nchunks > 1) load_idx(1);
  unsigned long long* const As64 = reinterpret_cast<unsigned long long*>(As);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();   // previous chunk's fragment reads are done
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      unsigned h0, m0, l0, h1, m1, l1;
      bf_split2(__uint_as_float(xv[q].x), __uint_as_float(xv[q].y), h0, m0, l0);
      bf_split2(__uint_as_float(xv[q].z), __uint_as_float(xv[q].w), h1, m1, l1);
      const int o = x_row[q] * (LDU * 2) + x_j[q];                    // 8-byte slots: plane p of the row starts at slot 8 p
      As64[o] = (unsigned long long)h0 | ((unsigned long long)h1 << 32);
      As64[o + 8] = (unsigned long long)m0 | ((unsigned long long)m1 << 32);
      As64[o + 16] = (unsigned long long)l0 | ((unsigned long long)l1 << 32);
    }
#pragma unroll
    for (int q = 0; q < NB; ++q)
      if (b_row[q] >= 0) Bs[b_row[q] * LDU + b_w[q]] = bv[q];
    __syncthreads();
    if (c + 1 < nchunks) {
      issue(c + 1);                                                   // (its indices arrived during the previous chunk)
      if (c + 2 < nchunks) load_idx(c + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[p][i] = __builtin_bit_cast(bf16x8, As[((wm * TM + i) * 32 + r31) * LDU + p * 4 + ks * 2 + half]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bf[p][j] = __builtin_bit_cast(bf16x8, Bs[((wn * TN + j) * 32 + r31) * LDU + p * 4 + ks * 2 + half]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {           // smallest terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
        }
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = colblock + (wn * TN + j) * 32 + r31;
    if (col >= a.cout) continue;
    const float b = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (r >= npos) continue;
        a.out[(size_t)(pos0 + r) * a.cout + col] = act1(acc[i][j][e] + b, a.act, a.slope);
      }
  }
}

// (Round 3, tools/gemm_h2_probe.py + PCC_DBG on the level-2 composite shape 58 051 x 128 x 21 952, and tools/write_probe.hip:
//  the chip stores this 5.1 GB buffer in 0.90 ms at best (5.65 TB/s, this kernel's own store pattern, any occupancy); this
//  kernel takes 1.68-1.78 = LDS skeleton 0.38 + loads 0.05 + MFMA 0.27 + stores 0.56 measured one at a time, but loads + stores
//  + skeleton = 1.36 = (loads + skeleton 0.43) + (stores + skeleton 0.94): L2 reads and HBM-bound stores of one CU do not
//  overlap, whatever issues them.  Built and measured against it, bit-identical results, all slower and removed: start-up skew
//  between the workgroups of a CU (no change); a persistent LDS-DMA chunk stream (global_load_lds into a 4-slot ring three
//  chunks ahead, operands stored in HBM in the LDS image, 16-byte stores after a quad transpose): 2.17 ms with every wave
//  loading and storing (vmcnt orders a wave's stores with its loads), 2.15 ms with four loader waves and eight store-only
//  compute waves, 2.08 with nt / write-through stores; its loads + stores alone take 2.0 ms.  DESIGN.md section 8.)
// The dense products in scaled fp16 pairs (see k_feat_split_h): the structure of k_gemm_bf2 with two planes per operand
// (8 units of 16 bytes per 32-channel piece, LDS rows of 9 units: 9 is odd, so a fragment read's 16 rows fall on 16 different
// bank quads), three MFMA terms, and the row and column scales applied to the accumulators on the way out.
// TN = 32-column MFMA tiles per wave: 2 -> the 128 x 128 workgroup tile, 4 -> 128 x 256 (round 4).  Per tile the kernel reads
// (128 + BN) operand rows of NCH * 128 B from L2 for 128 * BN * 4 B of products: 2 B read per B written at BN = 128, 1.5 at
// BN = 256 -- with the product stores out of the operands' way (non-temporal) the L2 -> LDS operand stream is what is left to
// shrink.  128 accumulator registers per lane, two workgroups per CU.
template <int NCH, int TN = 2>
__global__ void __launch_bounds__(256, TN == 2 ? 3 : 2) k_gemm_h2(ConvArgs a) {
  constexpr int BM = 128, BN = 64 * TN, LDU = 9, NB = BN / 32;
  constexpr unsigned ROWB = NCH * 128u;                // bytes of a feature row's planes
  __shared__ __attribute__((aligned(16))) uint4 As[BM * LDU];
  __shared__ __attribute__((aligned(16))) uint4 Bs[BN * LDU];
  __shared__ __attribute__((aligned(16))) float rs[BM];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int cpx = gridDim.x >> 3;
  const int wid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
  const int gy = (a.cout_pad + BN - 1) / BN;
  int tile_id, colblock;
  if (gy > 8) {
    const int g = wid / (8 * gy), rem = wid - g * 8 * gy;
    colblock = (rem >> 3) * BN;
    tile_id = g * 8 + (rem & 7);
  } else {
    tile_id = wid / gy;
    colblock = (wid - tile_id * gy) * BN;
  }
  const long long p0 = (long long)tile_id * BM;
  if (p0 >= a.n_out) return;
  const int npos = (int)min((long long)BM, a.n_out - p0);

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(a.feath) + (size_t)p0 * ROWB, (short)0, (int)((unsigned)npos * ROWB), 0x00020000);
  const float* const wplanes = a.wp + a.wp_elems + bf_plane_elems(a.wp_elems);      // fp16 planes behind the bf16 planes
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(wplanes) + (size_t)colblock * 128u;
  const float* const cinv = wplanes + a.wp_elems;                                    // [cout_pad] column 1/scale
  const unsigned b_stride = (unsigned)a.cout_pad * 128u;
  const int bcols = min(BN, a.cout_pad - colblock);                                  // (the last 256-wide block may hold 128 columns)
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(wb), (short)0, (int)((NCH - 1) * b_stride + (unsigned)bcols * 128u), 0x00020000);

  unsigned vA[4], ld[NB];
#pragma unroll
  for (int j = 0; j < 4; ++j) vA[j] = (unsigned)((j * 256 + tid) >> 3) * ROWB + (unsigned)(tid & 7) * 16u;
#pragma unroll
  for (int j = 0; j < NB; ++j) ld[j] = (unsigned)((j * 256 + tid) >> 3) * LDU + (unsigned)(tid & 7);
  // column (tid >> 3) + 32 j of the block; columns past cout_pad (second half of the last wide block) read zeros
  unsigned vB[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) vB[j] = ((tid >> 3) + 32 * j < bcols) ? (unsigned)tid * 16u + (unsigned)j * 4096u : BUF_OOB;

  uint4 av[4], bv[NB];
  auto issue = [&](int cbi) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      av[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, PCC_DBG_ON(a, 4) ? BUF_OOB : vA[j] + (unsigned)cbi * 128u, 0, 0));
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if constexpr (TN == 2) {
        bv[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, PCC_DBG_ON(a, 4) ? BUF_OOB : vB[0], (int)((unsigned)cbi * b_stride + (unsigned)j * 4096u), 0));
      } else {
        bv[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, (PCC_DBG_ON(a, 4) || vB[j] == BUF_OOB) ? BUF_OOB : vB[j] + (unsigned)cbi * b_stride, 0, 0));
      }
    }
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm = w >> 1, wn = w & 1;
  const int half = lane >> 5, r31 = lane & 31;
  const unsigned fa = (unsigned)((wm * 64 + r31) * LDU + half), fb = (unsigned)((wn * 32 * TN + r31) * LDU + half);

  issue(0);
  if (tid < BM) rs[tid] = tid < npos ? a.frow_inv[p0 + tid] : 0.f;
  float cs[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = colblock + wn * 32 * TN + j * 32 + r31;
    cs[j] = col < a.cout_pad ? cinv[col] : 0.f;
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    __syncthreads();   // previous chunk's fragment reads are done
#pragma unroll
    for (int j = 0; j < 4; ++j) As[ld[j]] = av[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) Bs[ld[j]] = bv[j];
    __syncthreads();
    if (c + 1 < NCH) issue(c + 1);            // next chunk's global loads fly during this chunk's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    if (PCC_DBG_ON(a, 2)) continue;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 af[2][2], bf[2][TN];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) af[p][i] = __builtin_bit_cast(f16x8, As[fa + i * 32 * LDU + p * 4 + ks * 2]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[p][j] = __builtin_bit_cast(f16x8, Bs[fb + j * 32 * LDU + p * 4 + ks * 2]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {           // small terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
        }
    }
  }
  if (PCC_DBG_ON(a, 1)) { if (acc[0][0][0] != 12345.678f) return; }

  // ---- stores: element e of acc[i][j] is row wm*64 + i*32 + (e&3) + 8*(e>>2) + 4*half, column wn*32*TN + j*32 + r31 of the tile
  const unsigned ncol = (unsigned)a.cout;
  float* const obase = a.out + (size_t)p0 * ncol + colblock;
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
      obase, (short)0, (int)(((unsigned)(npos - 1) * ncol + min((unsigned)BN, ncol - (unsigned)colblock)) * 4u), 0x00020000);
  const unsigned vO = ((unsigned)(wm * 64 + 4 * half) * ncol + (unsigned)(wn * 32 * TN + r31)) * 4u;
  const bool full = npos == BM && (unsigned)colblock + BN <= ncol;
  // The product buffer is written once and read back by the gather-sum long after it left the caches (5 GB per level):
  // non-temporal stores keep it from evicting the operands this kernel re-reads from L2 (round 3: 3.4 -> 4.2 TB/s of
  // algorithmic traffic on the composite levels, decode -0.6 ms; PCC_NT bit 0).
  // (32-deep products have hardly any operand to protect, and as a pure stream non-temporal stores are the slower ones --
  //  4.2 against 5.0 TB/s, tools/gemm_nt_probe.sh: the hint is taken from 64 input channels on; PCC_NT bit 6 forces it.)
  const bool nt = (a.nt & 1) != 0 && (NCH >= 2 || (a.nt & 64));
  const int row_lim = npos - wm * 64 - 4 * half;
  const int col_lim = (int)ncol - colblock - wn * 32 * TN - r31;
  float guard_mr = 0.f, guard_mc = 0.f;
#pragma unroll
  for (int j = 0; j < TN; ++j) guard_mc = fmaxf(guard_mc, cs[j]);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const float4 r4 = *reinterpret_cast<const float4*>(&rs[wm * 64 + i * 32 + 8 * e4 + 4 * half]);
      const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
      guard_mr = fmaxf(guard_mr, fmaxf(fmaxf(r4.x, r4.y), fmaxf(r4.z, r4.w)));
#pragma unroll
      for (int e1 = 0; e1 < 4; ++e1) {
        const int e = e4 * 4 + e1;
        const int rrow = i * 32 + e1 + 8 * e4;
        const unsigned so = (unsigned)rrow * ncol * 4u;      // scalar
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float v = acc[i][j][e] * (rr[e1] * cs[j]);
          if (full && nt) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsO, vO + (unsigned)j * 128u, (int)so, 2);
          } else if (full) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsO, vO + (unsigned)j * 128u, (int)so, 0);
          } else {                                           // last row tile / column block: invalid elements go out of range
            const unsigned off = (rrow < row_lim && j * 32 < col_lim) ? vO + (unsigned)j * 128u + so : BUF_OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsO, off, 0, 0);
          }
        }
      }
    }
  // range guard: elements of a row / column far below its maximum are carried with an ABSOLUTE error of
  // 2^-28 of that maximum, so a product's error can reach cin * 2^-27 * max|row| * max|column|; the scales bound the maxima
  // (max < 2^15 / scale).  A lane's rows x a lane's columns are exactly its outputs.
  if (a.guard && guard_mr * guard_mc * (8.f * (float)a.cin) > a.guard_lim) atomicOr(a.guard, 1);
}

// The gathered pair GEMM (pcc_conv_fwd_pairs, pcc_convt_fwd_rows: one kernel offset per 128-pair tile, T[pair] = x[in(pair)] W[k])
// in the scaled fp16 form of k_gemm_h2: a pair's product row is scaled like its input row, the weights per (offset, column).
template <int NCH>
__global__ void __launch_bounds__(256, 3) k_pair_h2(ConvArgs a) {
  constexpr int BM = 128, BN = 128, LDU = 9;
  constexpr unsigned ROWB = NCH * 128u;
  __shared__ __attribute__((aligned(16))) uint4 As[BM * LDU];
  __shared__ __attribute__((aligned(16))) uint4 Bs[BN * LDU];
  __shared__ __attribute__((aligned(16))) float rs[BM];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int cpx = gridDim.x >> 3;
  const int wid = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
  const int gy = a.cout_pad / BN;
  const int tile_id = wid / gy;
  const int colblock = (wid - tile_id * gy) * BN;
  if (tile_id >= *a.n_tiles) return;
  const long long p0 = (long long)tile_id * BM;
  const int kid = a.tile_k[tile_id];

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(a.feath), (short)0, (int)(unsigned)((size_t)a.n_in * ROWB), 0x00020000);
  const float* const wplanes = a.wp + a.wp_elems + bf_plane_elems(a.wp_elems);
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(wplanes) + ((size_t)kid * NCH * a.cout_pad + colblock) * 128u;
  const float* const cinv = wplanes + a.wp_elems + (size_t)kid * a.cout_pad;
  const unsigned b_stride = (unsigned)a.cout_pad * 128u;
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(wb), (short)0, (int)((NCH - 1) * b_stride + BN * 128u), 0x00020000);

  unsigned vA[4], ld[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned u = (unsigned)(j * 256 + tid), row = u >> 3, wu = u & 7u;
    const int g = a.pair_in[p0 + row];                                       // input row of the pair (-1: padding, reads zeros)
    vA[j] = g >= 0 ? (unsigned)g * ROWB + wu * 16u : BUF_OOB;
    ld[j] = row * LDU + wu;
  }
  const unsigned vB = (unsigned)tid * 16u;

  uint4 av[4], bv[4];
  auto issue = [&](int cbi) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      av[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vA[j] == BUF_OOB ? BUF_OOB : vA[j] + (unsigned)cbi * 128u, 0, 0));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bv[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, vB, (int)((unsigned)cbi * b_stride + (unsigned)j * 4096u), 0));
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm = w >> 1, wn = w & 1;
  const int half = lane >> 5, r31 = lane & 31;
  const unsigned fa = (unsigned)((wm * 64 + r31) * LDU + half), fb = (unsigned)((wn * 64 + r31) * LDU + half);

  issue(0);
  if (tid < BM) {
    const int g = a.pair_in[p0 + tid];
    rs[tid] = g >= 0 ? a.frow_inv[g] : 0.f;
  }
  float cs[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) cs[j] = cinv[colblock + wn * 64 + j * 32 + r31];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) As[ld[j]] = av[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) Bs[ld[j]] = bv[j];
    __syncthreads();
    if (c + 1 < NCH) issue(c + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 af[2][2], bf[2][2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) af[p][i] = __builtin_bit_cast(f16x8, As[fa + i * 32 * LDU + p * 4 + ks * 2]);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[p][j] = __builtin_bit_cast(f16x8, Bs[fb + j * 32 * LDU + p * 4 + ks * 2]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
        }
    }
  }
  // ---- stores: the tile's 128 product rows are consecutive rows of T (padding pairs included: they are zero)
  const unsigned ncol = (unsigned)a.cout;
  float* const obase = a.out + (size_t)p0 * ncol + colblock;
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
      obase, (short)0, (int)(((unsigned)(BM - 1) * ncol + min((unsigned)BN, ncol - (unsigned)colblock)) * 4u), 0x00020000);
  const unsigned vO = ((unsigned)(wm * 64 + 4 * half) * ncol + (unsigned)(wn * 64 + r31)) * 4u;
  const bool full = (unsigned)colblock + BN <= ncol;
  const int col_lim = (int)ncol - colblock - wn * 64 - r31;
  float guard_mr = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const float4 r4 = *reinterpret_cast<const float4*>(&rs[wm * 64 + i * 32 + 8 * e4 + 4 * half]);
      const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
      guard_mr = fmaxf(guard_mr, fmaxf(fmaxf(r4.x, r4.y), fmaxf(r4.z, r4.w)));
#pragma unroll
      for (int e1 = 0; e1 < 4; ++e1) {
        const int e = e4 * 4 + e1;
        const unsigned so = (unsigned)(i * 32 + e1 + 8 * e4) * ncol * 4u;      // scalar
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float v = acc[i][j][e] * (rr[e1] * cs[j]);
          if (full && (a.nt & 2)) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsO, vO + (unsigned)j * 128u, (int)so, 2);
          } else if (full) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsO, vO + (unsigned)j * 128u, (int)so, 0);
          } else {
            const unsigned off = (j * 32 < col_lim) ? vO + (unsigned)j * 128u + so : BUF_OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsO, off, 0, 0);
          }
        }
      }
    }
  if (a.guard && guard_mr * fmaxf(cs[0], cs[1]) * (8.f * (float)a.cin) > a.guard_lim) atomicOr(a.guard, 1);   // range guard, as in k_gemm_h2
}

// ------------------------------------------------------------------------------------------
// Persistent form of the split kernel for the GEMM-shaped launches -- dense [n, cin] x [cin, ncol] (generative transposed
// convolutions, 1x1 convolutions, GDN) and the gathered pair GEMMs (one kernel offset per 128-pair tile).  Their
// reduction is only cin deep (4 chunks at cin = 128): with one tile per workgroup the tile's first loads (full memory
// latency) and its drain were exposed on every tile, and the matrix pipe idled two thirds of the time (SQ counters,
// DESIGN.md section 8).  Here a workgroup walks a strided sequence of tiles as ONE chunk stream: the loads of the next
// tile's first chunk are in flight while the current tile's last chunk is multiplied and its accumulators are stored.
// Work ids are dealt in contiguous ranges per XCD (L2 locality as in k_conv_mfma); the dense form visits row tiles in
// groups of 8 per column block so that a block's weights are fetched once per group.
// ------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, int MODE>
__global__ void __launch_bounds__(256, 2) k_gemm_bf(ConvArgs a) {
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int LDU = 13;
  __shared__ __attribute__((aligned(16))) uint4 As[BM * LDU];
  __shared__ __attribute__((aligned(16))) uint4 Bs[BN * LDU];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bool pair_mode = (a.pair_in != nullptr);
  const int gy = a.cout_pad / BN;
  const long long n_tiles = pair_mode ? *a.n_tiles : (a.n_out + BM - 1) / BM;
  const bool groups = !pair_mode && gy > 8;
  const long long total = (groups ? (n_tiles + 7) / 8 * 8 : n_tiles) * gy;
  const int nwg_x = gridDim.x >> 3;
  const long long per = (total + 7) / 8;
  const long long lo = (long long)(blockIdx.x & 7) * per, hi = min(total, lo + per);

  constexpr int NA = (BM * 12 + 255) / 256, NB = (BN * 12 + 255) / 256;
  int a_row[NA], a_w[NA], b_row[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) { const int u = j * 256 + tid; a_row[j] = u / 12; a_w[j] = u - a_row[j] * 12; if (u >= BM * 12) a_row[j] = -1; }
#pragma unroll
  for (int j = 0; j < NB; ++j) { const int u = j * 256 + tid; b_row[j] = (u < BN * 12) ? u / 12 : -1; }

  const int wm = w / WN, wn = w % WN;
  const int half = lane >> 5, r31 = lane & 31;
  const unsigned row_bytes = (unsigned)a.cin * 6u;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(a.featb), (short)0, (int)(unsigned)((size_t)a.n_in * row_bytes), 0x00020000);
  const float* wb = a.wp + a.wp_elems;
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wb), (short)0, (int)(unsigned)((size_t)bf_plane_elems(a.wp_elems) * 4), 0x00020000);

  struct Work { long long id; int tile, colblock, pos0, npos, kid; };
  auto decode = [&](long long id, Work& wk) {           // first valid work item at or after `id` (stride nwg_x); id >= hi: none
    for (; id < hi; id += nwg_x) {
      int tile, cb;
      if (groups) { const long long g = id / (8 * gy); const int rem = (int)(id - g * 8 * gy); cb = rem >> 3; tile = (int)(g * 8 + (rem & 7)); }
      else { tile = (int)(id / gy); cb = (int)(id - (long long)tile * gy); }
      if (tile < n_tiles) {
        wk.tile = tile; wk.colblock = cb * BN; wk.pos0 = tile * BM;
        wk.npos = pair_mode ? BM : (int)min((long long)BM, a.n_out - (long long)tile * BM);
        wk.kid = pair_mode ? a.tile_k[tile] : 0;
        break;
      }
    }
    wk.id = id;
  };
  auto load_rows = [&](const Work& wk, int (&rows)[NA]) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int rc = min(max(a_row[j], 0), wk.npos - 1);        // tail rows repeat the tile's last row (never stored)
      rows[j] = pair_mode ? a.pair_in[wk.pos0 + rc] : (wk.pos0 + rc);
      if (a_row[j] < 0) rows[j] = -1;
    }
  };
  auto issue = [&](const Work& wk, int cbi, const int (&rows)[NA], uint4 (&av)[NA], uint4 (&bv)[NB]) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const unsigned off = rows[j] >= 0 ? (unsigned)rows[j] * row_bytes + (unsigned)cbi * 192u + (unsigned)a_w[j] * 16u : BUF_OOB;
      av[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
    }
    const unsigned wbase = (unsigned)((wk.kid * a.ppo + cbi) * a.cout_pad + wk.colblock) * 192u;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const unsigned off = b_row[j] >= 0 ? wbase + (unsigned)(j * 256 + tid) * 16u : BUF_OOB;
      bv[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, off, 0, 0));
    }
  };

  Work cur, nxt;
  decode(lo + (blockIdx.x >> 3), cur);
  if (cur.id >= hi) return;
  int rows_cur[NA], rows_nxt[NA];
  uint4 av[NA], bv[NB];
  load_rows(cur, rows_cur);
  issue(cur, 0, rows_cur, av, bv);
  int c = 0;                                             // chunk of `cur` whose data sits in av / bv
  const int ppo = a.ppo;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // what the chunk after (cur, c) is: the same tile's next piece, or the first piece of the next work item
  bool have_nxt = false;
  if (ppo == 1) { decode(cur.id + nwg_x, nxt); have_nxt = nxt.id < hi; if (have_nxt) load_rows(nxt, rows_nxt); }

  while (true) {
    __syncthreads();   // previous chunk's fragment reads are done
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if (a_row[j] >= 0) As[a_row[j] * LDU + a_w[j]] = av[j];
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (b_row[j] >= 0) Bs[b_row[j] * LDU + (j * 256 + tid) - b_row[j] * 12] = bv[j];
    __syncthreads();
    const bool last_piece = (c + 1 == ppo);
    // prefetch the following chunk (possibly of the next tile) so that it flies during this chunk's MFMAs
    if (!last_piece) {
      issue(cur, c + 1, rows_cur, av, bv);
      if (c + 2 == ppo) { decode(cur.id + nwg_x, nxt); have_nxt = nxt.id < hi; if (have_nxt) load_rows(nxt, rows_nxt); }
    } else if (have_nxt) {
      issue(nxt, 0, rows_nxt, av, bv);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[p][i] = __builtin_bit_cast(bf16x8, As[((wm * TM + i) * 32 + r31) * LDU + p * 4 + ks * 2 + half]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bf[p][j] = __builtin_bit_cast(bf16x8, Bs[((wn * TN + j) * 32 + r31) * LDU + p * 4 + ks * 2 + half]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {           // smallest terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
        }
    }
    if (!last_piece) { ++c; continue; }

    // ---- tile finished: bias, activation (or GDN), store; then move on to the prefetched tile ----------------------
    {
      const int pos0 = cur.pos0, npos = cur.npos, colblock = cur.colblock;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = colblock + (wn * TN + j) * 32 + r31;
        const float b = (a.bias && col < a.cout) ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int r = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            float v = acc[i][j][e] + b;
            acc[i][j][e] = 0.f;
            if (r >= npos || col >= a.cout) continue;
            const size_t o = (size_t)(pos0 + r) * a.cout + col;
            if (MODE == MODE_CONV) v = act1(v, a.act, a.slope);
            else { const float x = a.feat[o]; v = (MODE == MODE_GDN) ? x / v : x * v; }
            a.out[o] = v;
          }
        }
      }
    }
    if (!have_nxt) break;
    cur = nxt;
#pragma unroll
    for (int j = 0; j < NA; ++j) rows_cur[j] = rows_nxt[j];
    c = 0;
    have_nxt = false;
    if (ppo == 1) { decode(cur.id + nwg_x, nxt); have_nxt = nxt.id < hi; if (have_nxt) load_rows(nxt, rows_nxt); }
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
// a += x * m with m = 1 or 0 (one rounding, so m = 1 gives exactly a + x)
__device__ inline void thin_fma(float4& a, const float4 x, float m) { a.x = fmaf(x.x, m, a.x); a.y = fmaf(x.y, m, a.y); a.z = fmaf(x.z, m, a.z); a.w = fmaf(x.w, m, a.w); }
__device__ inline void thin_fma(float& a, const float x, float m) { a = fmaf(x, m, a); }
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
  // z-run kernel only: optional tile table (band order, pcc_band_tiles_build) and the fused 16 -> 1 head projection
  const int* tiles = nullptr; const int* n_tiles = nullptr;
  const float* w2 = nullptr;      // [27][cout] second convolution of an occupancy head (thin layout), PROJ variant
  float* t = nullptr;             // [27][n_out] projections t[k][i] = <relu(h_i), w2_k>, PROJ variant
};

// LDS image of the narrow-output weights: [K][CIN/4][16][4] -- k-quad major, then the 16 output columns, 4 channels
// each.  A ds_read_b128 is served in groups of 16 lanes and every group holds each column r16 exactly once (lanes
// {0-3,12-15,20-27}, ... of MI355X_MICROARCH.md's LDS table), so with the column as the fastest 16-byte index the 16
// lanes of a group always hit 16 different bank quads: conflict-free.  (The round-1 layout [K][16][CIN+4] put the
// k-quad in the low address bits: SQ_LDS_BANK_CONFLICT = 1/2 SQ_LDS_IDX_ACTIVE, profiles/r01_sq_counters_conv.txt.)
template <int CIN>
__device__ __forceinline__ const float* wave16_w(const float* wl_s, int kid, int kq, int r16) {
  return wl_s + ((kid * (CIN / 4) + kq) * 16 + r16) * 4;
}

template <int CIN>
__global__ void __launch_bounds__(512) k_conv_wave16(Wave16Args a) {
  constexpr int G = CIN / 16;
  constexpr int NW = 8;                                            // waves per workgroup
  extern __shared__ __attribute__((aligned(16))) float wl_s[];   // [K][CIN/4][16][4]
  for (int i = threadIdx.x; i < a.K * 16 * (CIN / 4); i += 512)
    reinterpret_cast<float4*>(wl_s)[i] = reinterpret_cast<const float4*>(a.wl)[i];
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
          const float4 w = *reinterpret_cast<const float4*>(wave16_w<CIN>(wl_s, kid, 4 * g + q, r16));
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

template <int CIN, bool PROJ>
__global__ void __launch_bounds__(512) k_conv_wave16z(Wave16Args a) {
  constexpr int G = CIN / 16;
  constexpr int NW = 8;
  extern __shared__ __attribute__((aligned(16))) float wl_s[];   // [27][CIN/4][16][4] (+ PROJ: per-wave 16x17 scratch)
  for (int i = threadIdx.x; i < 27 * 16 * (CIN / 4); i += 512)
    reinterpret_cast<float4*>(wl_s)[i] = reinterpret_cast<const float4*>(a.wl)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
  const unsigned spc = (unsigned)a.n_out;                          // one segment: positions = output rows
  // tiles of <= 16 consecutive rows: plain 16-row cuts, or the band-ordered table of pcc_band_tiles_build (rows of one
  // (x, y-band) run per tile, bands outermost: the dx = +-1 neighbours of a band's current x-slab then stay in the
  // XCD's L2 until that slab is processed itself)
  const unsigned total_tiles = a.tiles ? (unsigned)*a.n_tiles : (spc + 15) / 16;
  const unsigned cpx = gridDim.x >> 3;
  const unsigned per_xcd = (total_tiles + 7) / 8;
  const unsigned xcd_lo = (blockIdx.x & 7) * per_xcd;
  const unsigned xcd_hi = min(total_tiles, xcd_lo + per_xcd);
  const float* wl_lane = wl_s + r16 * 4 + q * 64;                  // wave16_w(kid, 4g+q, r16) = wl_lane + (kid*(CIN/4) + 4g) * 64

  // PROJ: B operand of the head's second convolution, out[o] = b2 + sum_k <relu(h[nbr_k(o)]), w2_k>, evaluated as
  // t[k][i] = <relu(h_i), w2_k> for the tile in registers (one more 16x16x4 MFMA block), so h never goes to memory
  float w2r[2][4];
  float* hs = nullptr;
  if constexpr (PROJ) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 16 * nt + r16, c = 4 * j + q;
        w2r[nt][j] = (k < 27 && c < a.cout) ? a.w2[k * a.cout + c] : 0.f;
      }
    hs = wl_s + 27 * 16 * CIN + (threadIdx.x >> 6) * (16 * 17);
  }

  // One (dx,dy) group of a 16-row tile: the dz=0 rows, and the dz=-+1 rows, each either the neighbouring lane's dz=0
  // row (mask k*) or loaded.  All rows come through buffer loads whose offset is out of range for an absent or
  // not-needed row: those lanes read 0 without touching memory, the number of loads in flight is fixed (exact
  // s_waitcnt distances; conditional loads made the compiler wait for the prefetch itself), and no branch is left.
  struct Grp { float4 c[G], m[G], p[G]; unsigned km, kp; };
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.feat), (short)0, (int)(unsigned)((size_t)a.n_in * CIN * 4), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;

  for (unsigned wt = xcd_lo + (blockIdx.x >> 3) * NW + (threadIdx.x >> 6); wt < xcd_hi; wt += cpx * NW) {
    unsigned pos0, npos;
    if (a.tiles) {
      const unsigned tw = (unsigned)a.tiles[wt];
      pos0 = tw & 0x07FFFFFFu; npos = (tw >> 27) + 1;
    } else {
      pos0 = wt * 16; npos = min(16u, spc - pos0);
    }
    const unsigned r = pos0 + min((unsigned)r16, npos - 1);        // tail rows repeat the tile's last row, never stored
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
      const float4 w = *reinterpret_cast<const float4*>(wl_lane + (slot * (CIN / 4) + 4 * g) * 64);
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
    const float b = (a.bias && r16 < a.cout) ? a.bias[r16] : 0.f;
    if constexpr (!PROJ) {
      if (r16 < a.cout) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned lr = 4 * q + e;
          if (lr < npos) a.out[(size_t)(pos0 + lr) * a.cout + r16] = act1(acc0[e] + acc1[e] + acc2[e] + acc3[e] + b, a.act, a.slope);
        }
      }
    } else {
      // h tile (activation applied) -> per-wave LDS scratch [row][17] -> A operand (lane = row, k = channel quad)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        hs[(4 * q + e) * 17 + r16] = r16 < a.cout ? act1(acc0[e] + acc1[e] + acc2[e] + acc3[e] + b, a.act, a.slope) : 0.f;
      __builtin_amdgcn_wave_barrier();      // same wave writes and reads: LDS executes a wave's operations in order
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
      // t^T tile = W2 (A: lane = offset, k = channel quad) x h^T (B: lane = row): rows end up across the lanes, so each
      // store instruction writes four 64-byte runs of consecutive rows instead of 64 scattered words
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float hv = hs[r16 * 17 + 4 * j + q];
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[0][j], hv, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w2r[1][j], hv, d1, 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();      // the next tile's scratch writes stay behind these reads
      // D: lane (row = r16, q) holds t[k = 4q+e (+16)][row]
      if ((unsigned)r16 < npos) {
        float* tp = a.t + (size_t)(4 * q) * spc + pos0 + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          tp[(size_t)e * spc] = d0[e];
          if (16 + 4 * q + e < 27) tp[(size_t)(16 + e) * spc] = d1[e];
        }
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
// MFMA weight images: the fp32 layout, followed (cin a multiple of 32) by the three bf16 planes of the split path
static int64_t mfma_packed_total(int64_t fp32_elems, int cin) { return cin % 32 == 0 ? fp32_elems + bf_plane_elems(fp32_elems) : fp32_elems; }
static int split_planes(float* packed, int64_t fp32_elems, int cin, hipStream_t s) {
  if (cin % 32 != 0) return PCC_OK;
  k_split_packed<<<(unsigned)pcc_cdiv(fp32_elems / 2, 256), 256, 0, s>>>(packed, fp32_elems / 2, (unsigned*)(packed + fp32_elems));
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}
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

// convolutions that run as gathered pair GEMMs (5x5x5 and wider, 128+ output channels): the pack also carries the scaled fp16
// planes and the 1/scale of every (offset, column), fp32 image | bf16 planes | fp16 planes | K*cout_pad scales
static bool conv_has_h(int K, int cin, int cout) {
  return K >= 64 && cin % 32 == 0 && cin <= 256 && cout % 4 == 0 && bn_for(cout) == 128;
}
extern "C" int64_t pcc_conv_packed_elems(int32_t K, int32_t cin, int32_t cout) {
  if (K <= 0 || cin <= 0 || cout <= 0) return 0;
  switch (conv_kind(K, cin, cout)) {
    case KIND_MFMA: {
      const int64_t base = (int64_t)K * cin * cout_pad_for(cout);
      return mfma_packed_total(base, cin) + (conv_has_h(K, cin, cout) ? base + (int64_t)K * cout_pad_for(cout) : 0);
    }
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

// The same for cin a multiple of 32 with the three bf16 planes of the split path written in the same pass (round 4: pack + split
// were two launches per weight and a training step packs ~45 weights), reading the source through strides: W'[k][ci][co] =
// W[kk * sk + ci * sci + co * sco], kk = K-1-k when `flip` -- the data gradient's transposed / offset-reversed kernels are
// packed straight from the parameter (no torch flip / permute / copy in front).  One thread = two consecutive channels.
__global__ void k_pack_mfma_split(const float* __restrict__ W, int K, int cin, int cout, int cout_pad, long long sk, long long sci,
                                  long long sco, int flip, float* __restrict__ out, unsigned* __restrict__ planes) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // pair index of the packed image
  const long long pairs = (long long)K * cin * cout_pad / 2;
  if (t >= pairs) return;
  const int within = (int)(t & 15) * 2;
  const long long q = t >> 4;                                                 // packed row = (piece, column)
  const int col = (int)(q % cout_pad);
  const long long piece = q / cout_pad;
  const int ppo = cin >> 5;
  const int kid = (int)(piece / ppo), cbi = (int)(piece % ppo);
  const int ci = (cbi << 5) + within;
  const int kk = flip ? K - 1 - kid : kid;
  float v0 = 0.f, v1 = 0.f;
  if (col < cout) {
    const float* w = W + kk * sk + col * sco;
    v0 = w[ci * sci];
    v1 = w[(ci + 1) * sci];
  }
  *reinterpret_cast<float2*>(out + 2 * t) = make_float2(v0, v1);
  unsigned h, m, l;
  bf_split2(v0, v1, h, m, l);
  unsigned* d = planes + q * 48 + (t & 15);
  d[0] = h; d[16] = m; d[32] = l;
}

// W [K][cin][cout] -> wave16 layout [K][cin/4][16][4] (k-quad major, 16 zero-padded output columns, 4 channels each:
// the LDS image of k_conv_wave16*, see wave16_w)
__global__ void k_pack_wave16(const float* __restrict__ W, int K, int cin, int cout, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)K * 16 * cin) return;
  const int c4 = (int)(t & 3);
  const int o = (int)((t >> 2) & 15);
  const int kq = (int)((t >> 6) % (cin / 4));
  const int k = (int)(t / ((long long)cin * 16));
  const int c = kq * 4 + c4;
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

static int pack_weights_impl(const float* W, int32_t K, int32_t cin, int32_t cout, long long sk, long long sci, long long sco,
                             int flip, float* packed, int64_t packed_cap, hipStream_t s);

extern "C" int pcc_conv_pack_weights(const float* W, int32_t K, int32_t cin, int32_t cout, float* packed,
                                     int64_t packed_cap, void* stream) {
  return pack_weights_impl(W, K, cin, cout, (long long)cin * cout, cout, 1, 0, packed, packed_cap, (hipStream_t)stream);
}

// The pack of W'[k][ci][co] = W[(flip ? K-1-k : k)][co][ci] when `transpose` (W stored [K][cout][cin]: the kernel of the
// convolution whose data gradient is being computed), else of W itself with the offsets reversed.
extern "C" int pcc_conv_pack_weights_ex(const float* W, int32_t K, int32_t cin, int32_t cout, int32_t transpose, int32_t flip,
                                        float* packed, int64_t packed_cap, void* stream) {
  return pack_weights_impl(W, K, cin, cout, (long long)cin * cout, transpose ? 1 : cout, transpose ? cin : 1, flip ? 1 : 0, packed,
                           packed_cap, (hipStream_t)stream);
}

static int pack_weights_impl(const float* W, int32_t K, int32_t cin, int32_t cout, long long sk, long long sci, long long sco,
                             int flip, float* packed, int64_t packed_cap, hipStream_t s) {
  PCC_REQUIRE(W && packed && K >= 1 && K <= MAXK && cin >= 1 && cout >= 1, "pcc_conv_pack_weights: bad arguments");
  const bool plain = sk == (long long)cin * cout && sci == cout && sco == 1 && !flip;
  const int64_t total = pcc_conv_packed_elems(K, cin, cout);
  if (packed_cap < total) {   // a buffer sized with another layout's query (round 1: GDN sized by the conv query) is refused
    pcc_set_error("pcc_conv_pack_weights: packed buffer holds %lld floats, the layout needs %lld", (long long)packed_cap, (long long)total);
    return PCC_EWS;
  }
  const unsigned g = (unsigned)pcc_cdiv(total > 0 ? total : 1, 256);
  switch (conv_kind(K, cin, cout)) {
    case KIND_MFMA: {
      const int64_t base = (int64_t)K * cin * cout_pad_for(cout);
      if (cin % 32 == 0) {
        k_pack_mfma_split<<<(unsigned)pcc_cdiv(base / 2, 256), 256, 0, s>>>(W, K, cin, cout, cout_pad_for(cout), sk, sci, sco, flip,
                                                                             packed, (unsigned*)(packed + base));
        PCC_LAUNCH_CHECK();
      } else {
        PCC_REQUIRE(plain, "pcc_conv_pack_weights_ex: transposed / reversed source needs cin a multiple of 32");
        k_pack_mfma<<<(unsigned)pcc_cdiv(base, 256), 256, 0, s>>>(W, K, cin, cout, cout_pad_for(cout), cb_log2_for(cin), packed);
        PCC_LAUNCH_CHECK();
      }
      if (conv_has_h(K, cin, cout)) {
        const int cp = cout_pad_for(cout);
        float* const planes = packed + base + bf_plane_elems(base);
        k_split_packed_h<<<dim3((unsigned)pcc_cdiv(cp, 128), (unsigned)K), 128, 0, s>>>(packed, cin >> 5, cp, (unsigned char*)planes, planes + base);
        PCC_LAUNCH_CHECK();
      }
      break;
    }
    case KIND_WAVE16:
      PCC_REQUIRE(plain, "pcc_conv_pack_weights_ex: transposed / reversed source is packed for the MFMA layout only");
      k_pack_wave16<<<g, 256, 0, s>>>(W, K, cin, cout, packed);
      break;
    case KIND_THIN_T: case KIND_THIN:
      PCC_REQUIRE(plain, "pcc_conv_pack_weights_ex: transposed / reversed source is packed for the MFMA layout only");
      k_pack_thin<<<g, 256, 0, s>>>(W, K, cin, cout, packed);
      break;
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
// which kernel form a timed launch took, with the work the library itself knows (dense products: rows x columns x depth;
// gathered forms report 0 -- their pair counts live on the device, the caller accounts them)
struct ProfRec { int form; double flops, bytes; };
static std::vector<ProfRec> g_prof_recs;
static ProfRec g_form = {PCC_FORM_OTHER, 0.0, 0.0};
static void prof_note(int form, double flops, double bytes) { g_form = {form, flops, bytes}; }
static void prof_push() {
  g_prof_recs.push_back(g_form);
  g_form = {PCC_FORM_OTHER, 0.0, 0.0};
  ++g_launches;
}

extern "C" int pcc_prof_enable(int32_t on) {
  g_prof_on = on != 0;
  g_ev_used = 0;
  g_launches = 0;
  g_prof_recs.clear();
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
  g_prof_recs.clear();
  return PCC_OK;
}

extern "C" int64_t pcc_prof_sequence(int32_t* h_forms, int64_t cap) {
  const int64_t n = (int64_t)g_prof_recs.size();
  for (int64_t i = 0; i < n && i < cap && h_forms; ++i) h_forms[i] = g_prof_recs[(size_t)i].form;
  return n;
}

extern "C" int pcc_prof_collect_forms(double* h_ms, int64_t* h_launches, double* h_flops, double* h_bytes) {
  PCC_REQUIRE(h_ms && h_launches && h_flops && h_bytes, "pcc_prof_collect_forms: NULL array");
  for (int f = 0; f < PCC_FORM_COUNT; ++f) { h_ms[f] = 0.0; h_launches[f] = 0; h_flops[f] = 0.0; h_bytes[f] = 0.0; }
  for (size_t i = 0; i + 1 < g_ev_used && i / 2 < g_prof_recs.size(); i += 2) {
    PCC_CHECK_HIP(hipEventSynchronize(g_ev_pool[i + 1]));
    float t = 0.f;
    PCC_CHECK_HIP(hipEventElapsedTime(&t, g_ev_pool[i], g_ev_pool[i + 1]));
    const ProfRec& r = g_prof_recs[i / 2];
    const int f = (r.form >= 0 && r.form < PCC_FORM_COUNT) ? r.form : PCC_FORM_OTHER;
    h_ms[f] += t; h_launches[f] += 1; h_flops[f] += r.flops; h_bytes[f] += r.bytes;
  }
  g_ev_used = 0;
  g_launches = 0;
  g_prof_recs.clear();
  return PCC_OK;
}

static bool g_mfma_buf = getenv("PCC_MFMA_BUF") ? atoi(getenv("PCC_MFMA_BUF")) != 0 : true;
// split path (fp32 products as six bf16 MFMA terms, k_conv_mfma_bf) unless the call asks for PCC_ARITH_F32 (fp32-input MFMA kernels);
// dense / pair products in scaled fp16 pairs (k_gemm_h2, k_pair_h2) only under PCC_ARITH_H3
static bool split_ok(const ConvArgs& a) {
  return a.arith != PCC_ARITH_F32 && g_mfma_buf && a.cb_log2 == 5 && a.n_in > 0 && a.n_in * a.cin * 6 <= BUF_MAX_BYTES && a.wp_elems > 0 &&
         bf_plane_elems(a.wp_elems) * 4 <= BUF_MAX_BYTES;
}

// out = act(bias + sum_s part[s]) in ascending s (fixed order: deterministic); 4 channels per thread
__global__ void __launch_bounds__(256) k_splitk_reduce(const float* __restrict__ part, int S, long long n4, int cout4,
                                                       const float* __restrict__ bias, int act, float slope, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n4) return;
  float4 v = reinterpret_cast<const float4*>(part)[t];
  for (int sidx = 1; sidx < S; ++sidx) {
    const float4 p = reinterpret_cast<const float4*>(part)[(long long)sidx * n4 + t];
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
  }
  if (bias) {
    const float4 b = reinterpret_cast<const float4*>(bias)[t % cout4];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  v.x = act1(v.x, act, slope); v.y = act1(v.y, act, slope); v.z = act1(v.z, act, slope); v.w = act1(v.w, act, slope);
  reinterpret_cast<float4*>(out)[t] = v;
}

static int g_splitk_chunks = getenv("PCC_SPLITK_CHUNKS") ? atoi(getenv("PCC_SPLITK_CHUNKS")) : 40;
static bool g_splitk = getenv("PCC_SPLITK") ? atoi(getenv("PCC_SPLITK")) != 0 : true;
static bool g_gemm_persistent = getenv("PCC_GEMM_PERSISTENT") ? atoi(getenv("PCC_GEMM_PERSISTENT")) != 0 : false;   // measured slower (2 workgroups per CU): off
static bool g_splitk_tiles = getenv("PCC_SPLITK_TILES") ? atoi(getenv("PCC_SPLITK_TILES")) != 0 : true;
static int g_gemm2 = getenv("PCC_GEMM2") ? atoi(getenv("PCC_GEMM2")) : 1;       // 0: general kernel, 1: stripped dense-GEMM kernel
static int g_dbg = getenv("PCC_DBG") ? atoi(getenv("PCC_DBG")) : 0;
// non-temporal accesses of the streamed multi-GB buffers (bit 0 dense products' stores, 1 pair products' stores, 2 gather-sum
// product loads, 3 gather-sum output stores, 4 projection-plane stores, 5 projection-plane gathers); env PCC_NT
static int g_nt = getenv("PCC_NT") ? atoi(getenv("PCC_NT")) : 1;

// persistent GEMM form (identity rows or pair lists; a.featb set): 2 workgroups per CU (the kernel needs ~200 VGPRs)
template <int MODE>
static int launch_gemm_bf(const ConvArgs& a, hipStream_t s) {
  int dev = 0, cus = 0;
  PCC_CHECK_HIP(hipGetDevice(&dev));
  PCC_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int bn = bn_for(a.cout);
  const long long gy = a.cout_pad / bn;
  const long long tiles128 = a.pair_in ? a.n_out / PAIR_BM_C : pcc_cdiv(a.n_out, 128);
  long long work = tiles128 * gy;
  long long grid = (long long)cus * 2;
  if (work < grid) grid = work;
  grid = (grid + 7) / 8 * 8;
  if (grid < 8) grid = 8;
  if (bn == 128) k_gemm_bf<2, 2, 2, 2, MODE><<<(unsigned)grid, 256, 0, s>>>(a);
  else if (bn == 64) k_gemm_bf<2, 2, 2, 1, MODE><<<(unsigned)grid, 256, 0, s>>>(a);
  else k_gemm_bf<4, 1, 1, 1, MODE><<<(unsigned)grid, 256, 0, s>>>(a);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

template <int MODE>
static int launch_mfma(const ConvArgs& a_in, int tiles_bound_extra, hipStream_t s) {
  ConvArgs a = a_in;
  const int bn = bn_for(a.cout);
  const long long gy = a.cout_pad / bn;
  const bool gemm_groups = !a.hdr && !a.pair_in && gy > 8;      // k_conv_mfma*: row tiles in groups of 8 (whole groups in the grid)
  auto tiles = [&](int bm) {
    const long long t = pcc_cdiv(a.n_out, bm) + tiles_bound_extra;
    return gemm_groups ? (t + 7) / 8 * 8 : t;
  };
  int ksplit_grid = 1;
  auto grid = [&](int bm) { return dim3((unsigned)((tiles(bm) * gy * ksplit_grid + 7) / 8 * 8)); };   // 1-D, multiple of 8 (XCD ranges)
  // few rows: shrink the row tile until the grid covers the 256 CUs about twice
  const long long want = 512;
  const bool buf = g_mfma_buf && a.n_in > 0 && a.n_in * a.cin * 4 <= BUF_MAX_BYTES && a.wp_elems > 0 &&
                   a.wp_elems * 4 <= BUF_MAX_BYTES;
  const bool split = split_ok(a);
  // few row tiles but a deep (offset x channel-block) reduction -- the 15 k-row / 4 k-row / 1 k-row layers of the hyper-prior:
  // a serial loop of 100-160 chunks at ~2 us per chunk on a handful of CUs.  Cut the reduction over ksplit workgroups
  // (partial tiles in the library scratch, summed in fixed order): the chain gets ksplit times shorter and the grid fills.
  int ksplit = 1;
  // (round 4: also the K = 1 products with a very deep reduction -- the data gradient of a generative transposed convolution is
  //  dT [n, 125 * cout] x Wflat^T: 500 chunks in one workgroup per 128 rows, 45 workgroups, 0.58 ms for 23 GFLOP)
  if (split && g_splitk && MODE == MODE_CONV && !a.pair_in && !a.rows && (a.cout & 3) == 0) {
    const int depth = (a.hdr ? 27 : 1) * a.ppo;        // chunks of a 3x3x3 map (the maps that reach here; 5x5x5 take the pair form)
    if (tiles(128) * gy < 256 && depth >= 64) {
      ksplit = depth / g_splitk_chunks;                // ~40 chunks per workgroup
      if (ksplit > 8) ksplit = 8;
      if (ksplit < 2) ksplit = 1;
    }
  }
  // dense products of the generative transposed convolutions whose pack carries fp16 planes: three-term fp16 form
  if (split && a.arith == PCC_ARITH_H3 && a.wh_ok && MODE == MODE_CONV && !a.hdr && !a.pair_in && !a.rows && !a.bias && a.act == 0 && ksplit == 1 &&
      bn == 128 && (tiles(128) * gy >= want || a.feath) && (size_t)128 * a.cout * 4 < (1ull << 31) &&
      (a.ppo == 1 || a.ppo == 2 || a.ppo == 4 || a.ppo == 6 || a.ppo == 8)) {      // (caller's planes: the caller chose the form)
    if (!a.feath) PCC_TRY(make_planes_h(a, s));
    a.dbg = g_dbg;
    a.nt = g_nt;
    prof_note(PCC_FORM_GEMM_H2, 2.0 * a.n_out * a.cin * a.cout, 4.0 * ((double)a.n_out * a.cin + (double)a.n_out * a.cout + (double)a.cin * a.cout));
    // 128 x 256 tiles for the wide products (the 7x7x7 composites: 5 488 / 21 952 columns): a quarter less operand traffic from
    // L2, bit-identical results -- and 2 % SLOWER on the benchmark's three levels (2.77 against 2.72 ms per step, round 4: two
    // workgroups per CU instead of three; the kernel is bound by its product stores, not by the operand stream).  Off; env
    // PCC_GEMM_WIDE=1 selects it (tests/test_gpu_map_conv.py runs it in a child process).
    static const bool wide_on = getenv("PCC_GEMM_WIDE") ? atoi(getenv("PCC_GEMM_WIDE")) != 0 : false;
    const long long gy2 = (a.cout_pad + 255) / 256;
    if (wide_on && (a.ppo == 4 || a.ppo == 2) && a.cout_pad >= 2048 && (size_t)128 * a.cout * 4 + 1024 < (1ull << 31)) {
      long long t2 = pcc_cdiv(a.n_out, 128);
      if (gy2 > 8) t2 = (t2 + 7) / 8 * 8;
      if (t2 * gy2 >= 1024) {
        const dim3 gw((unsigned)((t2 * gy2 + 7) / 8 * 8));
        if (a.ppo == 4) k_gemm_h2<4, 4><<<gw, 256, 0, s>>>(a); else k_gemm_h2<2, 4><<<gw, 256, 0, s>>>(a);
        PCC_LAUNCH_CHECK();
        return PCC_OK;
      }
    }
    const dim3 g2 = grid(128);
    switch (a.ppo) {
      case 1: k_gemm_h2<1><<<g2, 256, 0, s>>>(a); break;
      case 2: k_gemm_h2<2><<<g2, 256, 0, s>>>(a); break;
      case 4: k_gemm_h2<4><<<g2, 256, 0, s>>>(a); break;
      case 6: k_gemm_h2<6><<<g2, 256, 0, s>>>(a); break;
      default: k_gemm_h2<8><<<g2, 256, 0, s>>>(a); break;
    }
    PCC_LAUNCH_CHECK();
    return PCC_OK;
  }
  if (split && a.featb) {
    PCC_REQUIRE(ksplit == 1, "launch_mfma: caller planes with a split reduction");
  } else if (split) {
    const size_t plane_bytes = pcc_align_up((size_t)a.n_in * a.cin * 6);
    const size_t part_bytes = ksplit > 1 ? (size_t)ksplit * (size_t)a.n_out * a.cout * 4 : 0;
    void* p = nullptr;
    PCC_TRY(lib_scratch(plane_bytes + part_bytes, &p));
    const long long pairs = (long long)a.n_in * a.cin / 2;
    k_feat_split<<<(unsigned)pcc_cdiv(pairs, 256), 256, 0, s>>>(a.feat, pairs, a.cin / 2, MODE != MODE_CONV ? 1 : 0, (unsigned*)p);
    PCC_LAUNCH_CHECK();
    a.featb = (const unsigned char*)p;
    if (ksplit > 1) { a.ksplit = ksplit; a.part = (float*)((char*)p + plane_bytes); }
  }
  a.dbg = g_dbg;
  a.nt = g_nt;
  ksplit_grid = a.ksplit;
  if (split && !a.hdr && g_gemm_persistent) return launch_gemm_bf<MODE>(a, s);
  // plain dense products (generative transposed convolutions): the stripped GEMM kernel
  if (split && g_gemm2 && MODE == MODE_CONV && !a.hdr && !a.pair_in && !a.rows && !a.bias && a.act == 0 && a.ksplit == 1 &&
      bn == 128 && tiles(128) * gy >= want && (size_t)128 * a.cout * 4 < (1ull << 31)) {
    const dim3 g2 = grid(128);
    bool done = true;
    prof_note(PCC_FORM_GEMM_BF2, 2.0 * a.n_out * a.cin * a.cout, 4.0 * ((double)a.n_out * a.cin + (double)a.n_out * a.cout + (double)a.cin * a.cout));
    switch (a.ppo) {
      case 1: k_gemm_bf2<1><<<g2, 256, 0, s>>>(a); break;
      case 2: k_gemm_bf2<2><<<g2, 256, 0, s>>>(a); break;
      case 4: k_gemm_bf2<4><<<g2, 256, 0, s>>>(a); break;
      case 6: k_gemm_bf2<6><<<g2, 256, 0, s>>>(a); break;
      case 8: k_gemm_bf2<8><<<g2, 256, 0, s>>>(a); break;
      default: done = false;
    }
    if (done) { PCC_LAUNCH_CHECK(); return PCC_OK; }
  }
  prof_note(split ? PCC_FORM_CONV_BF : PCC_FORM_CONV_F32, (!a.hdr && !a.pair_in) ? 2.0 * a.n_out * a.cin * a.cout : 0.0, 0.0);
#define PCC_LAUNCH_MFMA(WM, WN, TM, TN, BMV)                                                     \
  do {                                                                                           \
    if (split) k_conv_mfma_bf<WM, WN, TM, TN, MODE><<<grid(BMV), 256, 0, s>>>(a);                \
    else if (buf) k_conv_mfma<WM, WN, TM, TN, MODE, true><<<grid(BMV), 256, 0, s>>>(a);          \
    else k_conv_mfma<WM, WN, TM, TN, MODE, false><<<grid(BMV), 256, 0, s>>>(a);                  \
  } while (0)
  // (a split reduction multiplies the grid: count it, so that split layers keep the large row tile and its weight reuse)
  const long long ksg = g_splitk_tiles ? a.ksplit : 1;
  if (bn == 128) {
    if (tiles(128) * gy * ksg >= want) PCC_LAUNCH_MFMA(2, 2, 2, 2, 128);
    else if (tiles(64) * gy * ksg >= want) PCC_LAUNCH_MFMA(2, 2, 1, 2, 64);
    else PCC_LAUNCH_MFMA(1, 4, 1, 1, 32);
  } else if (bn == 64) {
    if (tiles(128) * gy >= want) PCC_LAUNCH_MFMA(2, 2, 2, 1, 128);
    else PCC_LAUNCH_MFMA(2, 2, 1, 1, 64);
  } else PCC_LAUNCH_MFMA(4, 1, 1, 1, 128);
#undef PCC_LAUNCH_MFMA
  PCC_LAUNCH_CHECK();
  if (a.ksplit > 1) {
    const long long n4 = a.n_out * a.cout / 4;
    k_splitk_reduce<<<(unsigned)pcc_cdiv(n4, 256), 256, 0, s>>>(a.part, a.ksplit, n4, a.cout / 4, a.bias, a.act, a.slope, a.out);
    PCC_LAUNCH_CHECK();
  }
  return PCC_OK;
}

static bool g_wave16_zrun = getenv("PCC_WAVE16_ZRUN") ? atoi(getenv("PCC_WAVE16_ZRUN")) != 0 : true;

template <int CIN>
static int launch_wave16(const Wave16Args& a, hipStream_t s) {
  const size_t lds = (size_t)a.K * 16 * CIN * sizeof(float);
  int dev = 0;
  PCC_CHECK_HIP(hipGetDevice(&dev));
  static unsigned long long attr_set = 0;                         // one bit per device (hipFuncSetAttribute is per device)
  if (!(attr_set >> (dev & 63) & 1ull)) {
    PCC_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv_wave16<CIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    PCC_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv_wave16z<CIN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    PCC_CHECK_HIP(hipFuncSetAttribute((const void*)k_conv_wave16z<CIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_set |= 1ull << (dev & 63);
  }
  prof_note(PCC_FORM_WAVE16, 0.0, 0.0);
  const long long tiles = pcc_cdiv(a.n_out, 32) + (a.rows ? PCC_MAP_MAX_SEG : 0);
  long long want = pcc_cdiv(tiles, 8);
  want = (want + 7) / 8 * 8;                                     // multiple of 8: one contiguous tile range per XCD
  const unsigned grid = (unsigned)(want < 512 ? want : 512);     // persistent: 2 workgroups (16 waves) per CU re-use the LDS weights
  // 3x3x3 conv map in canonical row order (k_map_conv: one segment, all 27 offsets, no row list): z-run reuse variant
  if (g_wave16_zrun && a.K == 27 && a.hdr && !a.rows && a.n_out * 27 < (1ll << 31) &&
      a.n_in * CIN * 4 <= 0xFFFFFE00ll) {                          // 32-bit buffer offsets
    if (a.t) k_conv_wave16z<CIN, true><<<grid, 512, lds + 8 * 16 * 17 * sizeof(float), s>>>(a);
    else k_conv_wave16z<CIN, false><<<grid, 512, lds, s>>>(a);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
  }
  PCC_REQUIRE(!a.t, "pcc_conv_head_fwd: the fused head needs a canonical 3x3x3 map (one segment, no row list)");
  k_conv_wave16<CIN><<<grid, 512, lds, s>>>(a);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// The same projections on the matrix pipe, for wide inputs (cin 32 / 64) and at most 32 projections (the one-channel heads:
// 27): t^T = W2 x^T as v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand (its 32 rows = the projections k) and 32 feature
// rows as the B operand (its 32 columns), so that an accumulator register holds t[k][32 consecutive rows] across the lanes of a
// half wave: every store instruction writes two 128-byte runs of the k-major planes -- the layout the gather reads.  A lane
// (row r = lane & 31, half h) carries the channels h * CIN/2 ... of its row (CIN/8 16-byte loads, its half of the row,
// contiguous) and of its projection (CIN/2 registers, loaded once per wave); CIN/2 MFMAs per 32 rows.  The VALU form above
// spends 27 * CIN FMAs + 27 * CIN/4 broadcast LDS reads per row (24 TFLOP/s on the 64-channel heads: issue-bound).
template <int CIN>
__global__ void __launch_bounds__(256) k_thin_project_mfma(const float* __restrict__ feat, long long n_in,
                                                           const float* __restrict__ wt, int kc, float* __restrict__ t) {
  constexpr int HC = CIN / 2, NV = HC / 4;
  const int lane = threadIdx.x & 63, r31 = lane & 31, half = lane >> 5;
  const long long ntiles = (n_in + 31) / 32;
  float wa[HC];
#pragma unroll
  for (int c = 0; c < HC; ++c) wa[c] = r31 < kc ? wt[r31 * CIN + half * HC + c] : 0.f;
  const long long tstep = (long long)gridDim.x * 4;
  long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  auto load = [&](long long tl, float4 (&x)[NV]) {
    long long row = tl * 32 + r31;
    if (row >= n_in) row = n_in - 1;                       // tail rows repeat the last row (never stored)
    const float4* src = reinterpret_cast<const float4*>(feat + row * CIN + half * HC);
#pragma unroll
    for (int v = 0; v < NV; ++v) x[v] = src[v];
  };
  auto run = [&](long long tl, const float4 (&x)[NV]) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {                         // fixed order: channels ascending inside each half
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[4 * v + 0], x[v].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[4 * v + 1], x[v].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[4 * v + 2], x[v].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[4 * v + 3], x[v].w, acc, 0, 0, 0);
    }
    const long long row = tl * 32 + r31;
    if (row < n_in) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = (e & 3) + 8 * (e >> 2) + 4 * half;
        if (k < kc) t[(long long)k * n_in + row] = acc[e];
      }
    }
  };
  float4 xa[NV], xb[NV];
  load(tile, xa);
  for (;;) {                                               // two tiles per trip, the buffers swapping roles
    const long long t1 = tile + tstep;
    if (t1 < ntiles) load(t1, xb);
    run(tile, xa);
    if (t1 >= ntiles) break;
    const long long t2 = t1 + tstep;
    if (t2 < ntiles) load(t2, xa);
    run(t1, xb);
    if (t2 >= ntiles) break;
    tile = t2;
  }
}

static bool g_thin_mfma = getenv("PCC_THIN_MFMA") ? atoi(getenv("PCC_THIN_MFMA")) != 0 : true;

template <int CIN>
static int launch_project(const float* feat, int64_t n_in, const float* wt, int kc, float* t, hipStream_t s) {
  if constexpr (CIN >= 32) {
    if (g_thin_mfma && kc <= 32) {
      const long long tiles = pcc_cdiv(n_in, 32);
      long long grid = pcc_cdiv(tiles, 4 * 4);             // ~4 tiles per wave: the weight registers are loaded once per wave
      if (grid > 4096) grid = 4096;
      if (grid < 1) grid = 1;
      k_thin_project_mfma<CIN><<<(unsigned)grid, 256, 0, s>>>(feat, n_in, wt, kc, t);
      PCC_LAUNCH_CHECK();
      return PCC_OK;
    }
  }
  k_thin_project<CIN><<<(unsigned)pcc_cdiv(n_in, 256), 256, (size_t)kc * CIN * sizeof(float), s>>>(feat, n_in, wt, kc, t);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// 4-channel inputs: output rows from which the flattened form (k_conv_in4_bf) replaces the offset-by-offset kernel; negative = never
static long long g_in4_min_rows = getenv("PCC_IN4_MIN_ROWS") ? atoll(getenv("PCC_IN4_MIN_ROWS")) : 65536;
extern "C" int pcc_set_in4_min_rows(int64_t rows) { g_in4_min_rows = rows; return PCC_OK; }

extern "C" int pcc_conv_fwd(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w,
                            const float* bias, int32_t K, int32_t cout, const int32_t* hdr, const int32_t* nbr,
                            const int32_t* rows, int64_t n_out, float* out, int32_t act, float slope,
                            void* ws, size_t ws_bytes, int32_t arith, int32_t* d_guard, void* stream) {
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
    PCC_TRY(set_arith(a, arith, d_guard, "pcc_conv_fwd"));
    const bool in4 = g_in4_min_rows >= 0 && arith != PCC_ARITH_F32;
    // the input layer: 4 channels, K * 4 <= 512 flattened into one reduction axis (k_conv_in4_bf); plain conv maps of >= 64 k
    // output rows (one segment, canonical row order: what pcc_kernel_map_build makes for a non-transposed map)
    if (in4 && cin == 4 && K * 4 <= 512 && K > 1 && hdr && !rows && bn_for(cout) == 128 && n_out >= g_in4_min_rows &&
        n_in * 16 <= BUF_MAX_BYTES && n_out * (long long)K < (1ll << 31) && ((uintptr_t)feat_in & 15) == 0) {
      void* wpl = nullptr;
      PCC_TRY(lib_scratch_small((size_t)16 * a.cout_pad * 192, &wpl));
      k_in4_weight_planes<<<(unsigned)pcc_cdiv(16 * a.cout_pad * 16, 256), 256, 0, s>>>(packed_w, K, a.cout_pad, (unsigned*)wpl);
      const long long gy = a.cout_pad / 128;
      const unsigned grid = (unsigned)((pcc_cdiv(n_out, 128) * gy + 7) / 8 * 8);
      prof_note(PCC_FORM_CONV_BF, 0.0, 0.0);
      k_conv_in4_bf<2><<<grid, 256, 0, s>>>(a, (const unsigned*)wpl, K);
      PCC_LAUNCH_CHECK();
    } else
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
    prof_push();
  }
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// Band-ordered tile table for stencil kernels over large canonical sets (k_conv_wave16z).
// Canonical order is (x, y, z) with x slowest: the dx = +-1 neighbours of a row live one whole x-slab away, and on the
// 14.5 M-row level of the decoder three slabs of features (8 MB) do not fit an XCD's 4 MB L2, so every row was
// fetched from the fabric three times (round 1: 14.6 GB per launch for 1.86 GB of input).  Here the y range is cut into
// bands; the rows of one (band, x) pair are a contiguous run of the canonical order (found by two binary searches);
// tiles are cut inside the runs and numbered band-major, x ascending.  An XCD's contiguous tile range then sweeps
// x inside one band: the band's part of a slab (~0.3 MB) is still in L2 when it is needed again as dx = 0 and dx = -1.
// Tile word: row0 | (rows - 1) << 27.
// ------------------------------------------------------------------------------------------
__global__ void k_band_segments(const long long* __restrict__ keys, long long n, int lo_x, int nx, int lo_y, int ny, int ts,
                                int nbands, int band_h, int* __restrict__ seg_row0, int* __restrict__ seg_tiles) {
  const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (sidx >= nbands * nx) return;
  const int band = sidx / nx, xi = sidx - band * nx;
  const long long x = (long long)lo_x + (long long)xi * ts + PCC_BIAS;
  const int cy0 = band * band_h, cy1 = min(ny, cy0 + band_h);
  auto lower = [&](long long key) {
    long long lo = 0, hi = n;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  long long b = 0, e = 0;
  if (cy0 < cy1) {
    const long long y0 = (long long)lo_y + (long long)cy0 * ts + PCC_BIAS, y1 = (long long)lo_y + (long long)cy1 * ts + PCC_BIAS;
    b = lower((x << 32) | (y0 << 16));
    e = lower((x << 32) | (y1 << 16));
  }
  seg_row0[sidx] = (int)b;
  seg_tiles[sidx] = (int)((e - b + 15) / 16);
  seg_row0[nbands * nx + sidx] = (int)(e - b);       // second half of the array: rows of the run
}

__global__ void __launch_bounds__(1024) k_band_scan(const int* __restrict__ seg_tiles, int nseg, int* __restrict__ seg_tile0,
                                                    int* __restrict__ n_tiles) {
  __shared__ int part[1024];
  const int per = (nseg + 1023) / 1024;
  const int b = threadIdx.x * per, e = min(nseg, b + per);
  int sum = 0;
  for (int i = b; i < e; ++i) sum += seg_tiles[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - sum;
  for (int i = b; i < e; ++i) { seg_tile0[i] = run; run += seg_tiles[i]; }
  if (threadIdx.x == 1023) *n_tiles = part[1023];
}

__global__ void __launch_bounds__(256) k_band_fill(const int* __restrict__ seg_row0, const int* __restrict__ seg_rows,
                                                   const int* __restrict__ seg_tile0, int nseg, int* __restrict__ tiles) {
  const int sidx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sidx >= nseg) return;
  const int rows = seg_rows[sidx], r0 = seg_row0[sidx], t0 = seg_tile0[sidx];
  const int nt = (rows + 15) / 16;
  for (int j = threadIdx.x & 63; j < nt; j += 64) {
    const int cnt = min(16, rows - 16 * j);
    tiles[t0 + j] = (r0 + 16 * j) | ((cnt - 1) << 27);
  }
}

extern "C" int64_t pcc_band_tiles_cap(int64_t n, int32_t nx, int32_t nbands) { return n / 16 + (int64_t)nx * nbands + 16; }
extern "C" size_t pcc_band_tiles_ws_bytes(int32_t nx, int32_t nbands) { return pcc_align_up((size_t)nx * nbands * 4) * 4 + 256; }

extern "C" int pcc_band_tiles_build(const int64_t* keys, int64_t n, int32_t lo_x, int32_t nx, int32_t lo_y, int32_t ny,
                                    int32_t ts, int32_t nbands, int32_t* tiles, int64_t tiles_cap, int32_t* n_tiles,
                                    void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(keys && tiles && n_tiles && ws && n > 0 && n < (1ll << 27), "pcc_band_tiles_build: bad arguments (rows must stay below 2^27)");
  PCC_REQUIRE(nx >= 1 && ny >= 1 && ts >= 1 && nbands >= 1 && (int64_t)nx * nbands <= (1 << 20), "pcc_band_tiles_build: bad lattice");
  PCC_REQUIRE(tiles_cap >= pcc_band_tiles_cap(n, nx, nbands), "pcc_band_tiles_build: tile table too small (pcc_band_tiles_cap)");
  if (ws_bytes < pcc_band_tiles_ws_bytes(nx, nbands)) { pcc_set_error("pcc_band_tiles_build: workspace too small"); return PCC_EWS; }
  const int nseg = nx * nbands;
  const size_t st = pcc_align_up((size_t)nseg * 4);
  int* seg_row0 = (int*)ws;                              // [2][nseg]: first row, row count
  int* seg_tiles = (int*)((char*)ws + 2 * st);
  int* seg_tile0 = (int*)((char*)ws + 3 * st);
  PCC_REQUIRE(st >= (size_t)nseg * 4, "pcc_band_tiles_build: internal");
  const int band_h = (ny + nbands - 1) / nbands;
  // seg_row0 holds both arrays back to back (k_band_segments writes seg_row0[nseg + s] = rows): needs 2*nseg ints
  k_band_segments<<<(unsigned)pcc_cdiv(nseg, 256), 256, 0, s>>>((const long long*)keys, n, lo_x, nx, lo_y, ny, ts, nbands, band_h,
                                                                seg_row0, seg_tiles);
  PCC_LAUNCH_CHECK();
  k_band_scan<<<1, 1024, 0, s>>>(seg_tiles, nseg, seg_tile0, n_tiles);
  PCC_LAUNCH_CHECK();
  k_band_fill<<<(unsigned)pcc_cdiv(nseg, 4), 256, 0, s>>>(seg_row0, seg_row0 + nseg, seg_tile0, nseg, tiles);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// Occupancy head in one pass over the features (model/transforms.py:141-160, `predict_i`):
//   logits = conv_k3(relu(conv_k3(x; W0, b0)); W2, b2),   W0: cin -> cmid <= 16,  W2: cmid -> 1
// k_conv_wave16z<.., PROJ> evaluates the first convolution, the ReLU and the projections t[k][i] = <h_i, w2_k> tile by
// tile (h never reaches memory), k_thin_gather sums t through the same 3x3x3 map in ascending offset order (fixed
// order, deterministic; the projection runs on the MFMA, so the last bits differ from k_thin_project's VALU dot).
// ------------------------------------------------------------------------------------------
extern "C" int pcc_conv_head_supported(int32_t cin, int32_t cmid) {
  return (cmid > 4 && cmid <= 16 && conv_kind(27, cin, cmid) == KIND_WAVE16 && (size_t)27 * 16 * cin * 4 + 8 * 16 * 17 * 4 <= 64 * 1024) ? 1 : 0;
}
extern "C" size_t pcc_conv_head_ws_bytes(int64_t n) { return (size_t)27 * (size_t)(n > 0 ? n : 1) * sizeof(float) + 256; }

extern "C" int pcc_conv_head_fwd(const float* feat, int64_t n, int32_t cin, const float* packed_w0, const float* bias0,
                                 int32_t cmid, const float* w2, const float* bias2, const int32_t* hdr, const int32_t* nbr,
                                 const int32_t* tiles, const int32_t* n_tiles, float* logits, void* ws, size_t ws_bytes,
                                 void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(feat && packed_w0 && w2 && hdr && nbr && logits && ws, "pcc_conv_head_fwd: NULL array");
  PCC_REQUIRE(pcc_conv_head_supported(cin, cmid), "pcc_conv_head_fwd: unsupported shape cin=%d cmid=%d", cin, cmid);
  PCC_REQUIRE((tiles == nullptr) == (n_tiles == nullptr), "pcc_conv_head_fwd: tiles and n_tiles go together");
  PCC_REQUIRE(n * 27 < (1ll << 31) && n * cin * 4 <= 0xFFFFFE00ll && g_wave16_zrun, "pcc_conv_head_fwd: set too large for 32-bit offsets");
  if (ws_bytes < pcc_conv_head_ws_bytes(n)) { pcc_set_error("pcc_conv_head_fwd: workspace too small"); return PCC_EWS; }
  Wave16Args a;
  a.feat = feat; a.wl = packed_w0; a.bias = bias0; a.hdr = hdr; a.nbr = nbr; a.rows = nullptr; a.out = nullptr;
  a.n_out = n; a.n_in = n; a.K = 27; a.cout = cmid; a.act = PCC_ACT_RELU; a.slope = 0.f;
  a.tiles = tiles; a.n_tiles = n_tiles; a.w2 = w2; a.t = (float*)ws;
  hipEvent_t e0, e1;
  if (g_prof_on) PCC_TRY(prof_event(&e0, s));
  if (cin == 16) PCC_TRY(launch_wave16<16>(a, s));
  else if (cin == 32) PCC_TRY(launch_wave16<32>(a, s));
  else PCC_TRY(launch_wave16<64>(a, s));
  if (g_prof_on) {
    PCC_TRY(prof_event(&e1, s));
    prof_push();
  }
  ThinGatherArgs g;
  g.t = (const float*)ws; g.bias = bias2; g.hdr = hdr; g.nbr = nbr; g.rows = nullptr; g.out = logits; g.n_in = n; g.n_out = n;
  g.cout = 1; g.act = PCC_ACT_NONE; g.slope = 0.f;
  k_thin_gather<1><<<(unsigned)pcc_cdiv(n, 256), 256, 0, s>>>(g);
  PCC_LAUNCH_CHECK();
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

// LPR lanes per output row, 4 channels per lane and pass; pair rows of JB offsets loaded independently.
// (Round 3: a form that fetches a row's K position entries side by side, finds the present ones by ballot and walks only those
//  in batches of 8 product loads measured the same 117 us per launch: with 8 waves per SIMD the empty offsets' round trips are
//  hidden, the kernel runs at the rate its T reads allow -- 3.7 TB/s.  Not kept.)
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
                                  int32_t arith, int32_t* d_guard, void* stream) {
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
    PCC_TRY(set_arith(a, arith, d_guard, "pcc_conv_fwd_pairs"));
    hipEvent_t e0, e1;
    if (g_prof_on) PCC_TRY(prof_event(&e0, s));
    const int bn = bn_for(cout);
    const long long gy = a.cout_pad / bn;
    const dim3 grid((unsigned)((padded_pairs / PAIR_BM * gy + 7) / 8 * 8));
    const bool buf = g_mfma_buf && n_in * cin * 4 <= BUF_MAX_BYTES && a.wp_elems * 4 <= BUF_MAX_BYTES;
    const bool split = split_ok(a);
    const bool pair_h = split && a.arith == PCC_ARITH_H3 && conv_has_h(K, cin, cout) && (size_t)n_in * cin * 4 <= (size_t)BUF_MAX_BYTES &&
                        (a.ppo == 1 || a.ppo == 2 || a.ppo == 4 || a.ppo == 6 || a.ppo == 8);
    prof_note(pair_h ? PCC_FORM_PAIR_H2 : split ? PCC_FORM_PAIR_BF : PCC_FORM_CONV_F32, 0.0, 0.0);
    if (pair_h) {                                   // scaled fp16 pairs, three MFMA terms (k_pair_h2)
      PCC_TRY(make_planes_h(a, s));
      switch (a.ppo) {
        case 1: k_pair_h2<1><<<grid, 256, 0, s>>>(a); break;
        case 2: k_pair_h2<2><<<grid, 256, 0, s>>>(a); break;
        case 4: k_pair_h2<4><<<grid, 256, 0, s>>>(a); break;
        case 6: k_pair_h2<6><<<grid, 256, 0, s>>>(a); break;
        default: k_pair_h2<8><<<grid, 256, 0, s>>>(a); break;
      }
    } else if (split) PCC_TRY(make_planes(a, false, s));
    if (pair_h) {}
    else if (split && g_gemm_persistent) PCC_TRY(launch_gemm_bf<MODE_CONV>(a, s));
    else if (bn == 128) { if (split) k_conv_mfma_bf<2, 2, 2, 2, MODE_CONV><<<grid, 256, 0, s>>>(a); else if (buf) k_conv_mfma<2, 2, 2, 2, MODE_CONV, true><<<grid, 256, 0, s>>>(a); else k_conv_mfma<2, 2, 2, 2, MODE_CONV, false><<<grid, 256, 0, s>>>(a); }
    else if (bn == 64) { if (split) k_conv_mfma_bf<2, 2, 2, 1, MODE_CONV><<<grid, 256, 0, s>>>(a); else if (buf) k_conv_mfma<2, 2, 2, 1, MODE_CONV, true><<<grid, 256, 0, s>>>(a); else k_conv_mfma<2, 2, 2, 1, MODE_CONV, false><<<grid, 256, 0, s>>>(a); }
    else { if (split) k_conv_mfma_bf<4, 1, 1, 1, MODE_CONV><<<grid, 256, 0, s>>>(a); else if (buf) k_conv_mfma<4, 1, 1, 1, MODE_CONV, true><<<grid, 256, 0, s>>>(a); else k_conv_mfma<4, 1, 1, 1, MODE_CONV, false><<<grid, 256, 0, s>>>(a); }
    PCC_LAUNCH_CHECK();
    if (g_prof_on) {
      PCC_TRY(prof_event(&e1, s));
      prof_push();
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
                                  int32_t act, float slope, void* int_ws, size_t int_ws_bytes, int32_t arith, int32_t* d_guard,
                                  void* stream) {
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
    PCC_TRY(set_arith(a, arith, d_guard, "pcc_convt_fwd_rows"));
    hipEvent_t e0, e1;
    if (g_prof_on) PCC_TRY(prof_event(&e0, s));
    const int bn = bn_for(cout);
    const long long gy = a.cout_pad / bn;
    const dim3 grid((unsigned)((tiles_cap * gy + 7) / 8 * 8));
    const bool buf = g_mfma_buf && n_in * cin * 4 <= BUF_MAX_BYTES && a.wp_elems * 4 <= BUF_MAX_BYTES;
    const bool split = split_ok(a);
    const bool pair_h = split && a.arith == PCC_ARITH_H3 && conv_has_h(K, cin, cout) && (size_t)n_in * cin * 4 <= (size_t)BUF_MAX_BYTES &&
                        (a.ppo == 1 || a.ppo == 2 || a.ppo == 4 || a.ppo == 6 || a.ppo == 8);
    prof_note(pair_h ? PCC_FORM_PAIR_H2 : split ? PCC_FORM_PAIR_BF : PCC_FORM_CONV_F32, 0.0, 0.0);
    if (pair_h) {                                   // scaled fp16 pairs, three MFMA terms (k_pair_h2)
      PCC_TRY(make_planes_h(a, s));
      switch (a.ppo) {
        case 1: k_pair_h2<1><<<grid, 256, 0, s>>>(a); break;
        case 2: k_pair_h2<2><<<grid, 256, 0, s>>>(a); break;
        case 4: k_pair_h2<4><<<grid, 256, 0, s>>>(a); break;
        case 6: k_pair_h2<6><<<grid, 256, 0, s>>>(a); break;
        default: k_pair_h2<8><<<grid, 256, 0, s>>>(a); break;
      }
    } else if (split) PCC_TRY(make_planes(a, false, s));
    if (pair_h) {}
    else if (split && g_gemm_persistent) PCC_TRY(launch_gemm_bf<MODE_CONV>(a, s));
    else if (bn == 128) { if (split) k_conv_mfma_bf<2, 2, 2, 2, MODE_CONV><<<grid, 256, 0, s>>>(a); else if (buf) k_conv_mfma<2, 2, 2, 2, MODE_CONV, true><<<grid, 256, 0, s>>>(a); else k_conv_mfma<2, 2, 2, 2, MODE_CONV, false><<<grid, 256, 0, s>>>(a); }
    else if (bn == 64) { if (split) k_conv_mfma_bf<2, 2, 2, 1, MODE_CONV><<<grid, 256, 0, s>>>(a); else if (buf) k_conv_mfma<2, 2, 2, 1, MODE_CONV, true><<<grid, 256, 0, s>>>(a); else k_conv_mfma<2, 2, 2, 1, MODE_CONV, false><<<grid, 256, 0, s>>>(a); }
    else { if (split) k_conv_mfma_bf<4, 1, 1, 1, MODE_CONV><<<grid, 256, 0, s>>>(a); else if (buf) k_conv_mfma<4, 1, 1, 1, MODE_CONV, true><<<grid, 256, 0, s>>>(a); else k_conv_mfma<4, 1, 1, 1, MODE_CONV, false><<<grid, 256, 0, s>>>(a); }
    PCC_LAUNCH_CHECK();
    if (g_prof_on) {
      PCC_TRY(prof_event(&e1, s));
      prof_push();
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

// dense-product packs (cin a multiple of 32): fp32 image | three bf16 planes | two scaled fp16 planes | 1/scale per column
static bool convt_has_h(int cin) { return cin % 32 == 0 && cin <= 256; }
extern "C" int64_t pcc_convt_packed_elems(int32_t K, int32_t cin, int32_t cout) {
  if (K <= 0 || cin <= 0 || cout <= 0 || !mfma_ok(cin, K * cout)) return 0;
  const int64_t base = (int64_t)cin * cout_pad_for(K * cout);
  return mfma_packed_total(base, cin) + (convt_has_h(cin) ? base + cout_pad_for(K * cout) : 0);
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
  const int64_t base = (int64_t)cin * cout_pad_for(K * cout);
  k_pack_convt<<<(unsigned)pcc_cdiv(base, 256), 256, 0, s>>>(W, K, cin, cout, K * cout, cout_pad_for(K * cout),
                                                            cb_log2_for(cin), packed);
  PCC_LAUNCH_CHECK();
  PCC_TRY(split_planes(packed, base, cin, s));
  if (convt_has_h(cin)) {
    const int cp = cout_pad_for(K * cout);
    float* const planes = packed + base + bf_plane_elems(base);
    k_split_packed_h<<<(unsigned)pcc_cdiv(cp, 128), 128, 0, s>>>(packed, cin >> 5, cp, (unsigned char*)planes, planes + base);
    PCC_LAUNCH_CHECK();
  }
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
                             int64_t n_out, float* T, float* out, int32_t act, float slope, int32_t arith, int32_t* d_guard,
                             void* stream) {
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
  a.wh_ok = convt_has_h(cin);
  PCC_TRY(set_arith(a, arith, d_guard, "pcc_convt_fwd"));
  hipEvent_t e0, e1;
  if (g_prof_on) PCC_TRY(prof_event(&e0, s));
  PCC_TRY(launch_mfma<MODE_CONV>(a, 0, s));
  if (g_prof_on) {
    PCC_TRY(prof_event(&e1, s));
    prof_push();
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
// Subset sums of the per-neighbour constants: tab[j][m][c] = sum over the set bits b of m (ascending) of ex_bias[7j + b][c].
// A row's 27-bit neighbour mask then costs four table rows instead of a loop over its ~22 set bits (the loop was a third of the
// gather-sum's VALU instructions, and the kernel is VALU-bound: 6.6e8 wave instructions on the last level, SQ counters).
__global__ void k_presence_tables(const float* __restrict__ ex_bias, int cout, float* __restrict__ tab) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 512 * cout) return;
  const int c = t % cout, m = (t / cout) & 127, j = t / (128 * cout);
  float sum = 0.f;
  for (int b = 0; b < 7; ++b) {
    const int k = 7 * j + b;
    if (k < 27 && ((m >> b) & 1)) sum += ex_bias[k * cout + c];
  }
  tab[t] = sum;
}

// non-temporal accesses of HIP's vector structs (the builtins take native vector types)
typedef float f32x4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4* p) {
  const f32x4n v = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float nt_load(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void nt_store(const float4& v, float4* p) {
  const f32x4n w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<f32x4n*>(p));
}
__device__ __forceinline__ void nt_store(float v, float* p) { __builtin_nontemporal_store(v, p); }

struct GatherCsrArgs {
  const float* T; const float* bias; const int* first; const int* pair_ids;
  const int* wg_end = nullptr;                            // slotted lists (pcc_coords_expand_grid_csr_slots): end of the last row of every 256 rows
  float* out; long long n_out; int cout, act; float slope; int lpr_log2;
  const int* ex_nbr; const float* ex_bias; int ex_K;      // optional: + sum over the existing neighbours k of ex_bias[k]
  const float* ex_tab;                                    //   as subset-sum tables [4][128][cout] over 7+7+7+6 neighbour bits (k_presence_tables)
  PccGrid ex_grid; const long long* out_keys;             //   presence flags from a [K][n_out] table (ex_nbr) or the set's grid index
  int nt;                                                 // g_nt: 4 = non-temporal product loads, 8 = non-temporal output stores
};

template <int VEC, int JB>
__global__ void __launch_bounds__(256) k_convt_gather_csr(GatherCsrArgs a) {
  typedef typename ThinVec<VEC>::T VT;
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << a.lpr_log2;
  const int rpw = 64 >> a.lpr_log2;
  const long long o = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + (lane >> a.lpr_log2);
  const int cl = lane & (lpr - 1);
  if (o >= a.n_out) return;
  const int cvec = a.cout / VEC;
  const int t0 = a.first[o];
  const int t1 = (a.wg_end && ((o & 255) == 255 || o + 1 == a.n_out)) ? a.wg_end[o >> 8] : a.first[o + 1];
  // optional constant per existing neighbour (two fused affine layers): the lanes of the row's group fetch the ex_K presence
  // flags side by side and share them by ballot (one load per lane instead of ex_K dependent loads: the serial loop cost
  // 2.1 ms on the level-2 head in round 2)
  unsigned long long present = 0;
  // 3x3x3 presence straight from the output set's bitmap, the nine (dx, dy) columns dealt over the row's lanes.  Branch-free
  // (round 3): a lane's <= 3 columns are 64-bit windows that start at the 32-bit word of the column's first cell (the 3-bit z
  // field never straddles), absent columns re-read cell 0 and are masked -- all of a lane's loads are in flight together and
  // are consumed after the pair loop below.  (The loop form waited for each column's word in turn: three exposed L2 latencies
  // per row on the last level, where a row has four lanes.)
  unsigned pw_lo[3] = {0, 0, 0}, pw_hi[3] = {0, 0, 0};
  int psh[3] = {64, 64, 64}, pcol[3] = {0, 0, 0};
  int p_nz = 0, p_dz0 = 0;
  if (a.ex_grid.bits && lpr < 4) {                                  // (<= 8 channels: a row has one or two lanes, the loop form)
    unsigned m = pcc_grid_nbr27(a.ex_grid, a.out_keys[o], cl, lpr, nullptr);
    for (int d = lpr >> 1; d >= 1; d >>= 1) m |= __shfl_xor((int)m, d);
    present = m;
  } else if (a.ex_grid.bits) {
    const PccGrid& g = a.ex_grid;
    const long long key = a.out_keys[o];
    const int b = (int)(key >> 48);
    const int cx = (((int)((key >> 32) & 0xFFFF) - (int)PCC_BIAS - g.lo[0]) >> g.ts_log2);
    const int cy = (((int)((key >> 16) & 0xFFFF) - (int)PCC_BIAS - g.lo[1]) >> g.ts_log2);
    const int cz = (((int)(key & 0xFFFF) - (int)PCC_BIAS - g.lo[2]) >> g.ts_log2);
    const int z_lo = cz > 0 ? cz - 1 : 0, z_hi = cz + 1 < g.dims[2] ? cz + 1 : g.dims[2] - 1;
    p_nz = z_hi - z_lo + 1;
    p_dz0 = z_lo - cz + 1;
    const long long col_stride = g.dims[2], slab_stride = (long long)g.dims[1] * g.dims[2];
    const long long cell0 = (((long long)b * g.dims[0] + cx) * g.dims[1] + cy) * g.dims[2] + z_lo;
    const long long cells = (long long)g.nbatch * g.dims[0] * slab_stride;
    const long long last_dw = 2 * ((cells + 63) >> 6) - 2;
    const unsigned* const bits32 = reinterpret_cast<const unsigned*>(g.bits);
    const int step = lpr < 9 ? lpr : 9;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int c = cl + t * step;
      const int dx = c % 3 - 1, dy = c / 3 - 1;
      const int nx = cx + dx, ny = cy + dy;
      const bool ok = c < 9 && (lpr >= 9 ? t == 0 : true) && nx >= 0 && ny >= 0 && nx < g.dims[0] && ny < g.dims[1];
      const long long cell = ok ? cell0 + dx * slab_stride + dy * col_stride : 0ll;
      const long long dw = cell >> 5, dw2 = dw < last_dw ? dw : last_dw;
      psh[t] = ok ? (int)(cell & 31) + 32 * (int)(dw - dw2) : 64;
      pcol[t] = c;
      pw_lo[t] = bits32[dw2];
      pw_hi[t] = bits32[dw2 + 1];
    }
  } else if (a.ex_nbr) {
    for (int k0 = 0; k0 < a.ex_K; k0 += lpr) {
      const int k = k0 + cl;
      const bool v = k < a.ex_K && a.ex_nbr[(long long)k * a.n_out + o] >= 0;
      const unsigned long long bal = __ballot(v);
      present |= ((bal >> ((lane >> a.lpr_log2) << a.lpr_log2)) & (lpr == 64 ? ~0ull : ((1ull << lpr) - 1ull))) << k0;
    }
  }
  if (a.ex_grid.bits && lpr >= 4) {                                  // finish the presence mask from the windows fetched above
    unsigned pm = 0;
    const unsigned fmask = (1u << p_nz) - 1u;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const unsigned long long w = (unsigned long long)pw_lo[t] | ((unsigned long long)pw_hi[t] << 32);
      const unsigned f = psh[t] < 64 ? (unsigned)(w >> (psh[t] & 63)) & fmask : 0u;
      pm |= ((f & 1u) | ((f & 2u) << 8) | ((f & 4u) << 16)) << (pcol[t] + 9 * p_dz0);     // bit t of the field -> k = c + 9 (dz0 + t)
    }
    for (int d = lpr >> 1; d >= 1; d >>= 1) pm |= __shfl_xor((int)pm, d);
    present = pm;
  }
  for (int cv = cl; cv < cvec; cv += lpr) {
    VT acc;
    thin_zero(acc);
    // branch-free batches: slots past the end of the list re-read the last pair (same cache line) and are weighted 0, so the
    // JB index loads and then the JB product loads of a batch are independent and in flight together
    for (int t = t0; t < t1; t += JB) {
      int pid[JB];
#pragma unroll
      for (int u = 0; u < JB; ++u) pid[u] = a.pair_ids[min(t + u, t1 - 1)];
      VT x[JB];
      if (a.nt & 4) {
#pragma unroll
        for (int u = 0; u < JB; ++u) x[u] = nt_load(reinterpret_cast<const VT*>(a.T + (long long)pid[u] * a.cout) + cv);
      } else {
#pragma unroll
        for (int u = 0; u < JB; ++u) x[u] = reinterpret_cast<const VT*>(a.T + (long long)pid[u] * a.cout)[cv];
      }
#pragma unroll
      for (int u = 0; u < JB; ++u) thin_fma(acc, x[u], (t + u < t1) ? 1.f : 0.f);     // fixed order: pair id ascending
    }
    if (a.ex_tab) {                                                        // constants of the existing neighbours: four subset sums
      const VT* tb = reinterpret_cast<const VT*>(a.ex_tab);
      const unsigned m = (unsigned)present;
      thin_acc(acc, tb[(m & 127u) * cvec + cv]);
      thin_acc(acc, tb[(128u + ((m >> 7) & 127u)) * cvec + cv]);
      thin_acc(acc, tb[(256u + ((m >> 14) & 127u)) * cvec + cv]);
      thin_acc(acc, tb[(384u + ((m >> 21) & 63u)) * cvec + cv]);
    }
    VT b;
    thin_zero(b);
    if (a.bias) b = reinterpret_cast<const VT*>(a.bias)[cv];
    thin_acc(acc, b);
    thin_act(acc, a.act, a.slope);
    if (a.nt & 8) nt_store(acc, reinterpret_cast<VT*>(a.out + o * a.cout) + cv);
    else reinterpret_cast<VT*>(a.out + o * a.cout)[cv] = acc;
  }
}

// (Round 4 built the head's 27 projections INTO this kernel for the 16-channel level -- from the gather-sum's registers, on the
//  matrix pipe, hidden layer never stored -- three ways: stored straight from the MFMA layout (64-byte half lines per wave) 1.60 ms,
//  a wave making four passes to collect whole lines in registers 2.14 (a quarter of the occupancy), the workgroup's planes staged
//  through LDS 1.76 -- against 1.13 for this kernel + 0.58 for k_thin_project_z.  The gather-sum is latency-bound: every
//  instruction added behind its loads costs more than the streaming projection pass saves.  Removed; round-4 history.)
static int presence_tables(const float* ex_bias, int cout, const float** tab, hipStream_t s) {
  void* p = nullptr;
  PCC_TRY(lib_scratch_small((size_t)512 * cout * 4, &p));
  k_presence_tables<<<(unsigned)pcc_cdiv(512 * cout, 256), 256, 0, s>>>(ex_bias, cout, (float*)p);
  PCC_LAUNCH_CHECK();
  *tab = (const float*)p;
  return PCC_OK;
}

// ex_grid / ex_keys: presence source of pcc_convt_fwd_csr_grid (the output set's grid index) or NULL
static int convt_fwd_csr_impl(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w,
                              const float* bias, int32_t K, int32_t cout, const int32_t* first,
                              const int32_t* pair_ids, int64_t n_out, float* T, float* out, int32_t act, float slope,
                              const int32_t* ex_nbr, int32_t ex_K, const float* ex_bias, const PccGrid* ex_grid,
                              const long long* ex_keys, int32_t arith, int32_t* d_guard, void* stream, const int32_t* wg_end = nullptr) {
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
  a.wh_ok = convt_has_h(cin);
  PCC_TRY(set_arith(a, arith, d_guard, "pcc_convt_fwd_csr"));
  hipEvent_t e0, e1;
  if (g_prof_on) PCC_TRY(prof_event(&e0, s));
  PCC_TRY(launch_mfma<MODE_CONV>(a, 0, s));
  if (g_prof_on) {
    PCC_TRY(prof_event(&e1, s));
    prof_push();
  }
  GatherCsrArgs g;
  g.T = T; g.bias = bias; g.first = first; g.pair_ids = pair_ids; g.out = out; g.n_out = n_out; g.cout = cout;
  g.act = act; g.slope = slope; g.ex_nbr = ex_nbr; g.ex_bias = ex_bias; g.ex_K = ex_K; g.nt = g_nt;
  g.wg_end = wg_end;
  g.ex_grid.bits = nullptr; g.out_keys = nullptr;
  if (ex_grid) { g.ex_grid = *ex_grid; g.out_keys = ex_keys; g.ex_nbr = nullptr; }
  g.ex_tab = nullptr;
  if (ex_bias) {
    PCC_REQUIRE(ex_K == 27, "pcc_convt_fwd_csr: the per-neighbour constants are those of a 3x3x3 neighbourhood (ex_K=%d)", ex_K);
    PCC_TRY(presence_tables(ex_bias, cout, &g.ex_tab, s));
  }
  const int vec = (cout % 4 == 0) ? 4 : 1;
  int l = 0;
  while ((1 << l) < cout / vec && l < 6) ++l;
  g.lpr_log2 = l;
  const int64_t waves = pcc_cdiv(n_out, 64 >> l);
  const unsigned gg = (unsigned)pcc_cdiv(waves, 4);
  // pair slots per batch of independent loads: narrow outputs (the last level, ~4 pairs per row) take 4, the others 8
  // (measurement: the gather-sum of a composite level is event-timed too -- with the dense products it is the SURVEY 8d unit)
  const bool timed_gather = g_prof_on && ex_grid;
  hipEvent_t g0, g1;
  if (timed_gather) PCC_TRY(prof_event(&g0, s));
  // (round 4 probe: 8 slots on the last level as well -- 1.698 vs 1.703 ms per composite level: the gather-sum is bound by the
  //  memory system's rate on 64-byte pieces, not by loads in flight)
  if (vec == 4 && l <= 2) k_convt_gather_csr<4, 4><<<gg, 256, 0, s>>>(g);
  else if (vec == 4) k_convt_gather_csr<4, 8><<<gg, 256, 0, s>>>(g);
  else k_convt_gather_csr<1, 8><<<gg, 256, 0, s>>>(g);
  PCC_LAUNCH_CHECK();
  if (timed_gather) {
    PCC_TRY(prof_event(&g1, s));
    prof_note(PCC_FORM_GATHER_CSR, 0.0, 0.0);
    prof_push();
  }
  return PCC_OK;
}

extern "C" int pcc_convt_fwd_csr(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w,
                                 const float* bias, int32_t K, int32_t cout, const int32_t* first,
                                 const int32_t* pair_ids, int64_t n_out, float* T, float* out, int32_t act, float slope,
                                 const int32_t* ex_nbr, int32_t ex_K, const float* ex_bias, int32_t arith, int32_t* d_guard,
                                 void* stream) {
  return convt_fwd_csr_impl(feat_in, n_in, cin, packed_w, bias, K, cout, first, pair_ids, n_out, T, out, act, slope, ex_nbr,
                            ex_K, ex_bias, nullptr, nullptr, arith, d_guard, stream);
}

static int ilog2_i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static PccGrid grid_from_host(const uint64_t* bits, const int32_t* rank, const int32_t* h) {
  PccGrid g;
  g.bits = (const unsigned long long*)bits; g.rank = rank;
  for (int i = 0; i < 3; ++i) { g.lo[i] = h[i]; g.dims[i] = h[3 + i]; }
  g.ts_log2 = ilog2_i(h[6]); g.nbatch = h[7];
  return g;
}

// pcc_convt_fwd_csr with the constant-per-existing-neighbour term taken from the OUTPUT set's own grid index instead of a
// [27][n_out] neighbour table: the composite up+head convolutions then need no 3x3x3 kernel map of the candidate set at all
// (1.6 GB to write and 1.6 GB to read twice on the benchmark's last level).
extern "C" int pcc_convt_fwd_csr_grid(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w,
                                      const float* bias, int32_t K, int32_t cout, const int32_t* first,
                                      const int32_t* pair_ids, int64_t n_out, float* T, float* out, int32_t act, float slope,
                                      const int64_t* out_keys, const uint64_t* out_bits, const int32_t* out_rank,
                                      const int32_t* h_out, const float* ex_bias, const int32_t* wg_end, int32_t arith,
                                      int32_t* d_guard, void* stream) {
  PCC_REQUIRE(out_keys && out_bits && out_rank && h_out && ex_bias, "pcc_convt_fwd_csr_grid: NULL array");
  const PccGrid ex = grid_from_host(out_bits, out_rank, h_out);
  return convt_fwd_csr_impl(feat_in, n_in, cin, packed_w, bias, K, cout, first, pair_ids, n_out, T, out, act, slope,
                            nullptr, 27, ex_bias, &ex, (const long long*)out_keys, arith, d_guard, stream, wg_end);
}

// ---- chunked form of the CSR generative transposed convolution --------------------------------------------------------
// The per-pair products T[n_in][K][cout] of a composite 7x7x7 level are 5 GB -- written by the GEMM, read once by the ordered
// gather-sum.  Input rows are canonical (x-major), so a run of consecutive parents touches a contiguous run of children; the
// path is therefore cut into parent chunks whose products fit the 256 MiB Infinity Cache: GEMM chunk c -> T (one staging
// buffer, re-used by every chunk) -> gather-sum of the children chunk c reaches.  A child whose pair list straddles chunks
// carries its partial sum in `out`; pair ids ascend with the parent row, so every child still adds its pairs in list order
// and the result is bit-identical to the one-pass form.
__global__ void __launch_bounds__(256) k_chunk_ranges(const long long* __restrict__ in_keys, long long n_in,
                                                      const long long* __restrict__ out_keys, long long n_out,
                                                      long long chunk_rows, int n_chunks, int ts_out, int4* ranges) {
  auto lower = [&](long long q) {
    long long lo = 0, hi = n_out;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (out_keys[mid] < q) lo = mid + 1; else hi = mid;
    }
    return (int)lo;
  };
  const long long reach = 3ll * ts_out;                                  // 7-wide kernel: children within +-3 output pitches
  for (int c = threadIdx.x; c < n_chunks; c += blockDim.x) {
    const long long r0 = (long long)c * chunk_rows;
    const long long r1 = r0 + chunk_rows < n_in ? r0 + chunk_rows : n_in;
    const long long k0 = in_keys[r0], k1 = in_keys[r1 - 1];
    const long long x0 = (k0 >> 32) & 0xFFFF, x1 = ((k1 >> 32) & 0xFFFF) + reach + 1;
    const long long lo_key = (k0 & 0x7FFF000000000000ll) | ((x0 > reach ? x0 - reach : 0ll) << 32);
    const long long hi_key = (k1 & 0x7FFF000000000000ll) + (x1 << 32);  // + : a carry out of the x field moves on to the next batch
    ranges[c] = make_int4(lower(lo_key), c == n_chunks - 1 ? (int)n_out : lower(hi_key), 0, 0);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < n_chunks; c += blockDim.x) {
    const int prev_hi = c ? ranges[c - 1].y : 0;                          // rows below: owned (if their list is empty) by an earlier chunk
    ranges[c].z = prev_hi;
    if (prev_hi < ranges[c].x) ranges[c].x = prev_hi;
  }
}

template <int VEC>
__global__ void __launch_bounds__(256) k_convt_gather_csr_chunk(GatherCsrArgs a, const int4* __restrict__ range, int pid_lo,
                                                                int pid_hi) {
  typedef typename ThinVec<VEC>::T VT;
  constexpr int JB = 8;
  const int4 rg = *range;
  const int lane = threadIdx.x & 63;
  const int lpr = 1 << a.lpr_log2;
  const int rpw = 64 >> a.lpr_log2;
  const int cl = lane & (lpr - 1);
  const int cvec = a.cout / VEC;
  const long long per_block = 4ll * rpw;
  for (long long base = rg.x + (long long)blockIdx.x * per_block; base < rg.y; base += (long long)gridDim.x * per_block) {
    const long long o = base + (long long)(threadIdx.x >> 6) * rpw + (lane >> a.lpr_log2);
    if (o >= rg.y) continue;
    const int t0 = a.first[o], t1 = a.first[o + 1];
    bool has_earlier = false, is_last = true;
    if (t0 == t1) {
      if (o < rg.z) continue;                                             // empty list: finished by the chunk that owns the row
    } else {
      const int first_pid = a.pair_ids[t0], last_pid = a.pair_ids[t1 - 1];
      if (last_pid < pid_lo || first_pid >= pid_hi) continue;             // finished earlier / starts later
      has_earlier = first_pid < pid_lo;
      is_last = last_pid < pid_hi;
    }
    unsigned long long present = 0;
    if (is_last && a.ex_grid.bits) {                                      // (uniform over the row's lane group)
      unsigned m = pcc_grid_nbr27(a.ex_grid, a.out_keys[o], cl, lpr < 9 ? lpr : 9, nullptr);
      for (int d = lpr >> 1; d >= 1; d >>= 1) m |= __shfl_xor((int)m, d, lpr);
      present = m;
    }
    for (int cv = cl; cv < cvec; cv += lpr) {
      VT acc;
      thin_zero(acc);
      if (has_earlier) acc = reinterpret_cast<const VT*>(a.out + o * a.cout)[cv];
      for (int t = t0; t < t1; t += JB) {
        int pid[JB];
#pragma unroll
        for (int u = 0; u < JB; ++u) {
          const int p = (t + u < t1) ? a.pair_ids[t + u] : -1;
          pid[u] = (p >= pid_lo && p < pid_hi) ? p - pid_lo : -1;
        }
        VT x[JB];
#pragma unroll
        for (int u = 0; u < JB; ++u) {
          thin_zero(x[u]);
          if (pid[u] >= 0) x[u] = reinterpret_cast<const VT*>(a.T + (long long)pid[u] * a.cout)[cv];
        }
#pragma unroll
        for (int u = 0; u < JB; ++u) thin_acc(acc, x[u]);     // fixed order: pair id ascending, continued from the stored partial
      }
      if (is_last) {
        if (a.ex_tab) {
          const VT* tb = reinterpret_cast<const VT*>(a.ex_tab);
          const unsigned m = (unsigned)present;
          thin_acc(acc, tb[(m & 127u) * cvec + cv]);
          thin_acc(acc, tb[(128u + ((m >> 7) & 127u)) * cvec + cv]);
          thin_acc(acc, tb[(256u + ((m >> 14) & 127u)) * cvec + cv]);
          thin_acc(acc, tb[(384u + ((m >> 21) & 63u)) * cvec + cv]);
        }
        VT b;
        thin_zero(b);
        if (a.bias) b = reinterpret_cast<const VT*>(a.bias)[cv];
        thin_acc(acc, b);
        thin_act(acc, a.act, a.slope);
      }
      reinterpret_cast<VT*>(a.out + o * a.cout)[cv] = acc;
    }
  }
}

static long long g_chunk_bytes = getenv("PCC_T_CHUNK_MIB") ? atoll(getenv("PCC_T_CHUNK_MIB")) << 20 : 96ll << 20;
extern "C" int pcc_set_t_chunk_bytes(int64_t bytes) { g_chunk_bytes = bytes; return PCC_OK; }

static long long chunk_rows_for(int64_t n_in, int32_t K, int32_t cout) {
  long long rows = g_chunk_bytes / ((long long)K * cout * 4) / 1024 * 1024;      // whole groups of 8 row tiles of 128
  if (rows < 1024) rows = 1024;
  return rows < n_in ? rows : (n_in + 1023) / 1024 * 1024;
}

extern "C" size_t pcc_convt_chunk_t_bytes(int64_t n_in, int32_t K, int32_t cout) {
  return n_in <= 0 ? 256 : pcc_align_up((size_t)chunk_rows_for(n_in, K, cout) * K * cout * 4);
}
extern "C" size_t pcc_convt_chunk_ws_bytes(int64_t n_in, int32_t K, int32_t cout) {
  return n_in <= 0 ? 256 : pcc_align_up((size_t)pcc_cdiv(n_in, chunk_rows_for(n_in, K, cout)) * sizeof(int4));
}

extern "C" int pcc_convt_fwd_csr_chunked(const float* feat_in, int64_t n_in, int32_t cin, const float* packed_w,
                                         const float* bias, int32_t K, int32_t cout, const int32_t* first,
                                         const int32_t* pair_ids, int64_t n_out, const int64_t* in_keys,
                                         const int64_t* out_keys, int32_t ts_out, float* T, size_t t_bytes, float* out,
                                         int32_t act, float slope, const uint64_t* out_bits, const int32_t* out_rank,
                                         const int32_t* h_out, const float* ex_bias, void* ws, size_t ws_bytes, int32_t arith,
                                         int32_t* d_guard, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_out <= 0 || n_in <= 0) return PCC_OK;
  PCC_REQUIRE(feat_in && packed_w && first && pair_ids && in_keys && out_keys && T && out && ws,
              "pcc_convt_fwd_csr_chunked: NULL array");
  PCC_REQUIRE(K == 343, "pcc_convt_fwd_csr_chunked: 7x7x7 kernels only (K=%d)", K);
  PCC_REQUIRE(mfma_ok(cin, K * cout), "pcc_convt_fwd_csr_chunked: unsupported shape cin=%d cout=%d", cin, cout);
  PCC_REQUIRE(act >= 0 && act <= 2, "pcc_convt_fwd_csr_chunked: bad activation");
  PCC_REQUIRE(n_in * K < (1ll << 31) && n_out < (1ll << 31), "pcc_convt_fwd_csr_chunked: too many rows");
  PCC_REQUIRE(ts_out >= 1 && ts_out <= 16, "pcc_convt_fwd_csr_chunked: output pitch %d", ts_out);
  PCC_REQUIRE(!ex_bias || (out_bits && out_rank && h_out), "pcc_convt_fwd_csr_chunked: ex_bias needs the output set's grid index");
  const long long chunk_rows = chunk_rows_for(n_in, K, cout);
  const int n_chunks = (int)pcc_cdiv(n_in, chunk_rows);
  PCC_REQUIRE(t_bytes >= pcc_convt_chunk_t_bytes(n_in, K, cout) && ws_bytes >= pcc_convt_chunk_ws_bytes(n_in, K, cout),
              "pcc_convt_fwd_csr_chunked: staging buffer or workspace too small");
  int4* ranges = (int4*)ws;
  k_chunk_ranges<<<1, 256, 0, s>>>((const long long*)in_keys, n_in, (const long long*)out_keys, n_out, chunk_rows, n_chunks,
                                   ts_out, ranges);
  PCC_LAUNCH_CHECK();
  ConvArgs a;
  a.wp = packed_w; a.bias = nullptr; a.hdr = nullptr; a.nbr = nullptr; a.rows = nullptr; a.out = T;
  a.cin = cin; a.cout = K * cout; a.cout_pad = cout_pad_for(K * cout);
  a.wp_elems = (long long)cin * a.cout_pad;
  a.cb_log2 = cb_log2_for(cin); a.ppo = cin >> a.cb_log2; a.act = 0; a.slope = 0.f;
  a.feat = feat_in; a.n_in = n_in; a.n_out = n_in;
  PCC_TRY(set_arith(a, arith, d_guard, "pcc_convt_fwd_csr_chunked"));
  const bool split = split_ok(a);
  // the form the one-pass call would take for all rows (so that both give the same bits)
  const long long t128 = (pcc_cdiv(n_in, 128) + 7) / 8 * 8;
  const bool use_h = split && a.arith == PCC_ARITH_H3 && convt_has_h(cin) && (a.ppo == 1 || a.ppo == 2 || a.ppo == 4 || a.ppo == 6 || a.ppo == 8) &&
                     bn_for(a.cout) == 128 && t128 * (a.cout_pad / 128) >= 512;
  const unsigned char* planes = nullptr;
  const float* row_inv = nullptr;
  if (use_h) {                                                           // planes of every input row, once
    PCC_TRY(make_planes_h(a, s));
    planes = a.feath; row_inv = a.frow_inv;
    a.wh_ok = true;
  } else if (split) {
    PCC_TRY(make_planes(a, false, s));
    planes = a.featb;
  }
  GatherCsrArgs g;
  g.T = T; g.bias = bias; g.first = first; g.pair_ids = pair_ids; g.out = out; g.n_out = n_out; g.cout = cout;
  g.act = act; g.slope = slope; g.ex_nbr = nullptr; g.ex_bias = ex_bias; g.ex_K = 27;
  g.ex_grid.bits = nullptr; g.out_keys = nullptr;
  g.ex_tab = nullptr;
  if (ex_bias) {
    g.ex_grid = grid_from_host(out_bits, out_rank, h_out); g.out_keys = (const long long*)out_keys;
    PCC_TRY(presence_tables(ex_bias, cout, &g.ex_tab, s));
  }
  const int vec = (cout % 4 == 0) ? 4 : 1;
  int l = 0;
  while ((1 << l) < cout / vec && l < 6) ++l;
  g.lpr_log2 = l;
  int dev = 0, cus = 0;
  PCC_CHECK_HIP(hipGetDevice(&dev));
  PCC_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const unsigned ggrid = (unsigned)cus * 8;
  for (int c = 0; c < n_chunks; ++c) {
    const long long r0 = (long long)c * chunk_rows;
    const long long rows = r0 + chunk_rows < n_in ? chunk_rows : n_in - r0;
    a.feat = feat_in + r0 * cin; a.n_in = rows; a.n_out = rows;
    if (use_h) { a.feath = planes + (size_t)r0 * cin * 4; a.frow_inv = row_inv + r0; }
    else a.featb = planes ? planes + (size_t)r0 * cin * 6 : nullptr;
    hipEvent_t e0, e1;
    if (g_prof_on) PCC_TRY(prof_event(&e0, s));
    PCC_TRY(launch_mfma<MODE_CONV>(a, 0, s));
    if (g_prof_on) {
      PCC_TRY(prof_event(&e1, s));
      prof_push();
    }
    const int pid_lo = (int)(r0 * K), pid_hi = (int)((r0 + rows) * K);
    if (vec == 4) k_convt_gather_csr_chunk<4><<<ggrid, 256, 0, s>>>(g, ranges + c, pid_lo, pid_hi);
    else k_convt_gather_csr_chunk<1><<<ggrid, 256, 0, s>>>(g, ranges + c, pid_lo, pid_hi);
    PCC_LAUNCH_CHECK();
  }
  return PCC_OK;
}

// 3x3x3 convolution to <= 4 channels on a full set, neighbours from the set's grid index (no kernel map):
//   t[k*cout+co][i] = <feat[i], w_k[co]>  (k_thin_project),  out[o][co] = b + sum_k t[k*cout+co][nbr_k(o)]
static bool g_thin_grid1 = getenv("PCC_THIN_GRID1") ? atoi(getenv("PCC_THIN_GRID1")) != 0 : true;
struct ThinGridArgs {
  const float* t; const float* bias; const long long* keys; PccGrid g; float* out; long long n; int cout;
};

template <int COUT_MAX>
__global__ void __launch_bounds__(256) k_thin_gather_grid(ThinGridArgs a) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n) return;
  const PccGrid& g = a.g;
  const long long key = a.keys[p];
  const int b = (int)(key >> 48);
  const int cx = (((int)((key >> 32) & 0xFFFF) - (int)PCC_BIAS - g.lo[0]) >> g.ts_log2);
  const int cy = (((int)((key >> 16) & 0xFFFF) - (int)PCC_BIAS - g.lo[1]) >> g.ts_log2);
  const int cz = (((int)(key & 0xFFFF) - (int)PCC_BIAS - g.lo[2]) >> g.ts_log2);
  const int z_lo = cz > 0 ? cz - 1 : 0, z_hi = cz + 1 < g.dims[2] ? cz + 1 : g.dims[2] - 1;
  const int nz = z_hi - z_lo + 1;
  float acc[COUT_MAX];
#pragma unroll
  for (int o = 0; o < COUT_MAX; ++o) acc[o] = 0.f;
  // fixed order: (dx,dy) columns ascending, z ascending inside a column (no neighbour table: rows come from the bitmap + rank)
#pragma unroll
  for (int c = 0; c < 9; ++c) {
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
    int r = g.rank[wi] + __popcll(w0 & ((1ull << sh) - 1ull));
    while (f) {
      const int t = __ffs((int)f) - 1;
      f &= f - 1;
      const int k = c + 9 * (z_lo + t - cz + 1);
#pragma unroll
      for (int o = 0; o < COUT_MAX; ++o)
        if (o < a.cout) acc[o] += a.t[(long long)(k * a.cout + o) * a.n + r];
      ++r;
    }
  }
#pragma unroll
  for (int o = 0; o < COUT_MAX; ++o)
    if (o < a.cout) a.out[p * a.cout + o] = acc[o] + (a.bias ? a.bias[o] : 0.f);
}

// one output channel, branch-free: the 9 (bitmap word, rank) pairs of a row are fetched together, then its 27 projected
// values with buffer loads whose offset is out of range for an absent neighbour (27 independent loads in flight per row,
// where the loop form above serialised column after column behind its branches).  t must stay below 4 GB.
__global__ void __launch_bounds__(256) k_thin_gather_grid1(ThinGridArgs a) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n) return;
  const PccGrid& g = a.g;
  const long long key = a.keys[p];
  const int b = (int)(key >> 48);
  const int cx = (((int)((key >> 32) & 0xFFFF) - (int)PCC_BIAS - g.lo[0]) >> g.ts_log2);
  const int cy = (((int)((key >> 16) & 0xFFFF) - (int)PCC_BIAS - g.lo[1]) >> g.ts_log2);
  const int cz = (((int)(key & 0xFFFF) - (int)PCC_BIAS - g.lo[2]) >> g.ts_log2);
  const int z_lo = cz > 0 ? cz - 1 : 0, z_hi = cz + 1 < g.dims[2] ? cz + 1 : g.dims[2] - 1;
  const int nz = z_hi - z_lo + 1;
  const int dz0 = z_lo - cz + 1;
  unsigned long long w0[9], w1[9];
  int rk[9], sh[9];
  bool ok[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    const int nx = cx + c % 3 - 1, ny = cy + c / 3 - 1;
    ok[c] = !(nx < 0 || ny < 0 || nx >= g.dims[0] || ny >= g.dims[1]);
    const long long cell = ok[c] ? (((long long)b * g.dims[0] + nx) * g.dims[1] + ny) * g.dims[2] + z_lo : 0ll;
    const long long wi = cell >> 6;
    sh[c] = (int)(cell & 63);
    w0[c] = g.bits[wi];
    rk[c] = g.rank[wi];
    w1[c] = (sh[c] + nz > 64) ? g.bits[wi + 1] : 0ull;        // (rare: the field straddles two words)
  }
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.t), (short)0,
                                                                       (int)(unsigned)((size_t)27 * a.n * 4), 0x00020000);
  float v[27];
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    unsigned long long f64 = w0[c] >> sh[c];
    if (sh[c] + nz > 64) f64 |= w1[c] << (64 - sh[c]);
    const unsigned f = ok[c] ? ((unsigned)f64 & ((1u << nz) - 1u)) : 0u;
    const int r = rk[c] + __popcll(w0[c] & ((1ull << sh[c]) - 1ull));
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int k = c + 9 * (dz0 + t);                         // (k < 27 whenever bit t can be set: t < nz)
      const unsigned row = (unsigned)(r + __popc(f & ((1u << t) - 1u)));
      const unsigned off = ((f >> t) & 1u) ? ((unsigned)k * (unsigned)a.n + row) * 4u : BUF_OOB;
      v[c * 3 + t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsT, off, 0, 0));
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 27; ++i) acc += v[i];                    // fixed order: columns ascending, z ascending (absent: + 0)
  a.out[p] = acc + (a.bias ? a.bias[0] : 0.f);
}

// z-folded planes (round 3, one output channel over <= 16 hidden channels: the last level's head).  Canonical order is z fastest,
// so the dz = -1 / +1 neighbours of row i inside a (x, y) column are rows i - 1 / i + 1.  The projection pass therefore pre-adds a
// column's three terms for the row in the MIDDLE:   S_g[i] = <h_i, w(g,0)> + [i-1 adjacent] <h_{i-1}, w(g,-1)> + [i+1 adjacent]
// <h_{i+1}, w(g,+1)>   (g = the 9 (dx, dy) columns), and keeps the dz = -1 / +1 single terms as U_g[i], D_g[i] for the rare
// column whose middle cell is absent.  The gather then reads ONE value per column (9 x 4 B per row instead of 27 x 4 B, the
// same 27 planes in memory): 2.0 -> ~1.0 GB of L2 / HBM reads on the last level.
template <int CIN>
__global__ void __launch_bounds__(256) k_thin_project_z(const float* __restrict__ feat, const long long* __restrict__ keys,
                                                        long long n_in, long long ts, const float* __restrict__ wt,
                                                        float* __restrict__ t) {
  // A workgroup owns 256 consecutive rows (aligned 256-byte store runs per wave and plane).  Pass 1 projects every row on the
  // 27 kernels into LDS (one column per row + one halo column each side: the rows just outside the workgroup are projected by
  // 18 of its threads); pass 2 adds a column's neighbour terms from the adjacent LDS columns.  The loops over the kernels stay
  // rolled: unrolled, the compiler keeps all 27 x CIN weights in registers (256 VGPRs + spills, one wave per SIMD: 1.6 ms).
  extern __shared__ __attribute__((aligned(16))) float w_s[];          // 27 * CIN weights
  __shared__ float sd[27][258];                                        // [kernel][1 + thread] (+ halo columns 0 and 257)
  for (int i = threadIdx.x; i < 27 * CIN; i += 256) w_s[i] = wt[i];
  __syncthreads();
  const long long base = (long long)blockIdx.x * 256;
  const long long i = base + threadIdx.x;
  const bool valid = i < n_in;
  float4 x[CIN / 4];
#pragma unroll
  for (int c = 0; c < CIN / 4; ++c) x[c] = valid ? reinterpret_cast<const float4*>(feat + i * CIN)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  const long long key = valid ? keys[i] : -(1ll << 62);
  const bool adjm = valid && i > 0 && keys[i - 1] == key - ts;         // row i - 1 is the z - 1 cell of the same column
  const bool adjp = valid && i + 1 < n_in && keys[i + 1] == key + ts;
#pragma unroll 1
  for (int k = 0; k < 27; ++k) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < CIN / 4; ++c) {
      const float4 w = reinterpret_cast<const float4*>(w_s + k * CIN)[c];   // wave-uniform address: LDS broadcast
      acc += x[c].x * w.x + x[c].y * w.y + x[c].z * w.z + x[c].w * w.w;
    }
    sd[k][1 + threadIdx.x] = acc;
  }
  // the rows just outside the workgroup: thread e < 9 projects row base - 1 on w(e, dz = -1), thread 9 + e row base + 256 on
  // w(e, dz = +1) (same dot, same order of additions as above)
  if (threadIdx.x < 18) {
    const int e = threadIdx.x < 9 ? threadIdx.x : threadIdx.x - 9;
    const long long r = threadIdx.x < 9 ? base - 1 : base + 256;
    const int k = threadIdx.x < 9 ? e : e + 18;
    float acc = 0.f;
    if (r >= 0 && r < n_in) {
#pragma unroll
      for (int c = 0; c < CIN / 4; ++c) {
        const float4 xv = reinterpret_cast<const float4*>(feat + r * CIN)[c];
        const float4 w = reinterpret_cast<const float4*>(w_s + k * CIN)[c];
        acc += xv.x * w.x + xv.y * w.y + xv.z * w.z + xv.w * w.w;
      }
    }
    sd[k][threadIdx.x < 9 ? 0 : 257] = acc;
  }
  __syncthreads();
  if (!valid) return;
  const int col = 1 + threadIdx.x;
#pragma unroll 1
  for (int g = 0; g < 9; ++g) {
    const float lo = sd[g][col], mid = sd[g + 9][col], hi = sd[g + 18][col];
    const float from_dn = sd[g][col - 1];                               // <h_{i-1}, w(g, dz = -1)>
    const float from_up = sd[g + 18][col + 1];                          // <h_{i+1}, w(g, dz = +1)>
    t[(long long)g * n_in + i] = (mid + (adjm ? from_dn : 0.f)) + (adjp ? from_up : 0.f);
    t[(long long)(9 + g) * n_in + i] = lo;
    t[(long long)(18 + g) * n_in + i] = hi;
  }
}

// gather over the z-folded planes: per column the middle cell's S value, or -- middle absent -- the U / D singles of the cells
// below / above it.  Three buffer loads per column, at most two of them in range.
__global__ void __launch_bounds__(256) k_thin_gather_grid1z(ThinGridArgs a) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.n) return;
  const PccGrid& g = a.g;
  const long long key = a.keys[p];
  const int b = (int)(key >> 48);
  const int cx = (((int)((key >> 32) & 0xFFFF) - (int)PCC_BIAS - g.lo[0]) >> g.ts_log2);
  const int cy = (((int)((key >> 16) & 0xFFFF) - (int)PCC_BIAS - g.lo[1]) >> g.ts_log2);
  const int cz = (((int)(key & 0xFFFF) - (int)PCC_BIAS - g.lo[2]) >> g.ts_log2);
  const int z_lo = cz > 0 ? cz - 1 : 0, z_hi = cz + 1 < g.dims[2] ? cz + 1 : g.dims[2] - 1;
  const int nz = z_hi - z_lo + 1;
  const int dz0 = z_lo - cz + 1;                                      // dz index (0, 1, 2 = -1, 0, +1) of the field's bit 0
  unsigned long long w0[9], w1[9];
  int rk[9], sh[9];
  bool ok[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    const int nx = cx + c % 3 - 1, ny = cy + c / 3 - 1;
    ok[c] = !(nx < 0 || ny < 0 || nx >= g.dims[0] || ny >= g.dims[1]);
    const long long cell = ok[c] ? (((long long)b * g.dims[0] + nx) * g.dims[1] + ny) * g.dims[2] + z_lo : 0ll;
    const long long wi = cell >> 6;
    sh[c] = (int)(cell & 63);
    w0[c] = g.bits[wi];
    rk[c] = g.rank[wi];
    w1[c] = (sh[c] + nz > 64) ? g.bits[wi + 1] : 0ull;
  }
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.t), (short)0,
                                                                       (int)(unsigned)((size_t)27 * a.n * 4), 0x00020000);
  const int tm = 1 - dz0;                                             // bit of the middle cell (dz0 <= 1: it is inside the field)
  float v[27];
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    unsigned long long f64 = w0[c] >> sh[c];
    if (sh[c] + nz > 64) f64 |= w1[c] << (64 - sh[c]);
    const unsigned f = ok[c] ? ((unsigned)f64 & ((1u << nz) - 1u)) : 0u;
    const unsigned r = (unsigned)(rk[c] + __popcll(w0[c] & ((1ull << sh[c]) - 1ull)));
    const bool mid = (f >> tm) & 1u;
    const bool low = dz0 == 0 && (f & 1u);                            // the dz = -1 cell is bit 0, present only when z_lo = cz - 1
    const int tu = 2 - dz0;                                           // bit of the dz = +1 cell (may lie past the field: then absent)
    const bool upp = tu < nz && ((f >> tu) & 1u);
    const unsigned row_mid = r + __popc(f & ((1u << tm) - 1u));
    const unsigned row_up = r + __popc(f & ((1u << tu) - 1u));
    const unsigned n = (unsigned)a.n;
    const unsigned o_s = mid ? ((unsigned)c * n + row_mid) * 4u : BUF_OOB;
    const unsigned o_u = (!mid && low) ? ((unsigned)(9 + c) * n + r) * 4u : BUF_OOB;
    const unsigned o_d = (!mid && upp) ? ((unsigned)(18 + c) * n + row_up) * 4u : BUF_OOB;
    v[c * 3 + 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsT, o_s, 0, 0));
    v[c * 3 + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsT, o_u, 0, 0));
    v[c * 3 + 2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsT, o_d, 0, 0));
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 27; ++i) acc += v[i];                    // fixed order: columns ascending (absent: + 0)
  a.out[p] = acc + (a.bias ? a.bias[0] : 0.f);
}

// (Round 3 built a one-pass form of this convolution -- gather the 27 neighbours' hidden rows and dot them with w2 in
//  registers, dz = +-1 terms taken from the adjacent candidate by lane shuffle -- three times: columns walked one after the
//  other (latency-bound, +0.8 ms per step), all loads independent with index arithmetic per lane (issue-bound, +1.6 ms), index
//  arithmetic once per row and four lanes per row for coalesced 64-byte loads (+1.1 ms: 1.59 ms on the last level against
//  0.55 + 0.60 for project + gather).  Moving 9 x 64 B per output through L1 costs more than writing 27 floats per row and
//  gathering 27 x 4 B: the two-kernel form stays.)
// one-channel heads over 16 hidden channels: rows from which the z-folded planes are used (negative: never)
static long long g_thin_z_min_rows = getenv("PCC_THIN_Z_MIN_ROWS") ? atoll(getenv("PCC_THIN_Z_MIN_ROWS")) : (1ll << 20);
extern "C" int pcc_set_thin_z_min_rows(int64_t rows) { g_thin_z_min_rows = rows; return PCC_OK; }

extern "C" size_t pcc_thin_grid_ws_bytes(int64_t n, int32_t cout) { return (size_t)27 * cout * (size_t)(n > 0 ? n : 1) * sizeof(float) + 256; }

extern "C" int pcc_conv_thin_grid_fwd(const float* feat, int64_t n, int32_t cin, const float* packed_w /*thin layout [27][cout][cin]*/,
                                      const float* bias, int32_t cout, const int64_t* keys, const uint64_t* bits,
                                      const int32_t* rank, const int32_t* h_grid, float* out, void* ws, size_t ws_bytes,
                                      void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(feat && packed_w && keys && bits && rank && h_grid && out && ws, "pcc_conv_thin_grid_fwd: NULL array");
  PCC_REQUIRE(cout >= 1 && cout <= 4 && conv_kind(27, cin, cout) == KIND_THIN_T, "pcc_conv_thin_grid_fwd: unsupported shape cin=%d cout=%d", cin, cout);
  if (ws_bytes < pcc_thin_grid_ws_bytes(n, cout)) { pcc_set_error("pcc_conv_thin_grid_fwd: workspace too small"); return PCC_EWS; }
  float* t = (float*)ws;
  const int kc = 27 * cout;
  if (g_thin_z_min_rows >= 0 && cout == 1 && cin == 16 && (size_t)27 * n * 4 <= (size_t)BUF_MAX_BYTES && n >= g_thin_z_min_rows) {
    // narrow hidden layer over a large set (the last level): z-folded planes, one value per column in the gather
    k_thin_project_z<16><<<(unsigned)pcc_cdiv(n, 256), 256, (size_t)27 * 16 * sizeof(float), s>>>(
        feat, (const long long*)keys, n, (long long)h_grid[6], packed_w, t);
    ThinGridArgs az;
    az.t = t; az.bias = bias; az.keys = (const long long*)keys; az.g = grid_from_host(bits, rank, h_grid); az.out = out; az.n = n; az.cout = 1;
    k_thin_gather_grid1z<<<(unsigned)pcc_cdiv(n, 256), 256, 0, s>>>(az);
    PCC_LAUNCH_CHECK();
    return PCC_OK;
  }
  switch (cin) {
    case 4: PCC_TRY(launch_project<4>(feat, n, packed_w, kc, t, s)); break;
    case 8: PCC_TRY(launch_project<8>(feat, n, packed_w, kc, t, s)); break;
    case 16: PCC_TRY(launch_project<16>(feat, n, packed_w, kc, t, s)); break;
    case 32: PCC_TRY(launch_project<32>(feat, n, packed_w, kc, t, s)); break;
    default: PCC_TRY(launch_project<64>(feat, n, packed_w, kc, t, s)); break;
  }
  ThinGridArgs a;
  a.t = t; a.bias = bias; a.keys = (const long long*)keys; a.g = grid_from_host(bits, rank, h_grid); a.out = out; a.n = n; a.cout = cout;
  if (cout == 1 && (size_t)27 * n * 4 <= (size_t)BUF_MAX_BYTES && g_thin_grid1) k_thin_gather_grid1<<<(unsigned)pcc_cdiv(n, 256), 256, 0, s>>>(a);
  else if (cout == 1) k_thin_gather_grid<1><<<(unsigned)pcc_cdiv(n, 256), 256, 0, s>>>(a);
  else k_thin_gather_grid<4><<<(unsigned)pcc_cdiv(n, 256), 256, 0, s>>>(a);
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

extern "C" int64_t pcc_gdn_packed_elems(int32_t c) { return mfma_ok(c, c) ? mfma_packed_total((int64_t)c * cout_pad_for(c), c) : 0; }

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
  const int64_t base = (int64_t)c * cout_pad_for(c);
  k_gdn_pack<<<(unsigned)pcc_cdiv(base, 256), 256, 0, s>>>(beta_raw, gamma_raw, c, beta_bound, gamma_bound,
                                                          (float)pedestal, cout_pad_for(c), cb_log2_for(c), packed,
                                                          beta_eff);
  PCC_LAUNCH_CHECK();
  return split_planes(packed, base, c, s);
}

extern "C" int pcc_gdn_fwd(const float* x, int64_t n, int32_t c, const float* packed, const float* beta_eff,
                           int32_t inverse, float* out, int32_t arith, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(x && packed && beta_eff && out && x != out, "pcc_gdn_fwd: bad arguments");
  PCC_REQUIRE(mfma_ok(c, c), "pcc_gdn: channel count %d unsupported", c);
  ConvArgs a;
  a.feat = x; a.wp = packed; a.bias = beta_eff; a.hdr = nullptr; a.nbr = nullptr; a.rows = nullptr; a.out = out;
  a.n_out = n; a.cin = c; a.cout = c; a.cout_pad = cout_pad_for(c);
  a.n_in = n; a.wp_elems = (long long)c * a.cout_pad;
  a.cb_log2 = cb_log2_for(c); a.ppo = c >> a.cb_log2; a.act = 0; a.slope = 0.f;
  PCC_TRY(set_arith(a, arith, nullptr, "pcc_gdn_fwd"));
  // large sets: split folded into the staging (k_gdn_bf), no plane round trip; small ones keep the general kernel (its
  // smaller row tiles fill the chip better below ~30 k rows)
  static const bool fused = getenv("PCC_GDN_FUSED") ? atoi(getenv("PCC_GDN_FUSED")) != 0 : true;
  if (fused && split_ok(a) && c % 32 == 0 && (c & 3) == 0 && ((uintptr_t)x & 15) == 0 && n >= 32768) {
    const int bn = bn_for(c);
    if (bn == 128) {
      const long long gy = a.cout_pad / 128;
      const bool big = pcc_cdiv(n, 128) * gy >= 1024;
      const long long tiles = pcc_cdiv(n, big ? 128 : 64);
      const unsigned grid = (unsigned)((tiles * gy + 7) / 8 * 8);
      if (big) { if (inverse) k_gdn_bf<2, MODE_IGDN><<<grid, 256, 0, s>>>(a); else k_gdn_bf<2, MODE_GDN><<<grid, 256, 0, s>>>(a); }
      else { if (inverse) k_gdn_bf<1, MODE_IGDN><<<grid, 256, 0, s>>>(a); else k_gdn_bf<1, MODE_GDN><<<grid, 256, 0, s>>>(a); }
      PCC_LAUNCH_CHECK();
      return PCC_OK;
    }
  }
  if (inverse) return launch_mfma<MODE_IGDN>(a, 0, s);
  return launch_mfma<MODE_GDN>(a, 0, s);
}
