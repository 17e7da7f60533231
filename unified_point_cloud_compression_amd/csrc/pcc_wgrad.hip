// Weight gradient of the sparse convolution (BASELINE config 4: training step forward + backward; reference
// train.py:221-227 back-propagates through every ME.MinkowskiConvolution / GenerativeConvolutionTranspose):
//     dW[k][ci][co] = sum over the pairs (i,o) of offset k of  X[i][ci] * G[o][co]
// An MFMA GEMM per offset whose REDUCTION dimension is the pair list: M = ci, N = co, K = pairs.  A workgroup owns
// one (offset slot, 128x128 tile of dW, slice of positions); it gathers 32 pairs at a time (the input row and the
// output-gradient row of each pair) into LDS and feeds v_mfma_f32_32x32x2_f32 with transposed operand reads.
// Slices write partial tiles that a second kernel sums in slice order: deterministic, no atomics.
// The data gradient needs no kernel of its own: it is the forward kernel on the inverse map with W^T (sparse.py).
#include <stdlib.h>

#include "pcc_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int WG_PAIRS = 32;         // pairs staged per step
static constexpr int WG_SLICE = 4096;       // positions per workgroup slice

// 4 consecutive channels [col, col+4) of a row of `width` floats; 16-byte load when the row pitch allows it,
// guarded scalar loads for the thin layers (1 / 3 channels) and at the tile edge
__device__ inline float4 load4(const float* __restrict__ row, int width, int col) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((width & 3) == 0 && col + 3 < width) return *reinterpret_cast<const float4*>(row + col);
  if (col < width) v.x = row[col];
  if (col + 1 < width) v.y = row[col + 1];
  if (col + 2 < width) v.z = row[col + 2];
  if (col + 3 < width) v.w = row[col + 3];
  return v;
}

// the same for rows whose pitch is a multiple of 4 floats, WITHOUT a branch: an absent pair (row < 0) or a column beyond the
// row reads element 0 and is zeroed by a select.  (With load4 inside `cond ? load : 0` the compiler built a branch per load and
// waited for each: 527 branches and 54 `s_waitcnt vmcnt(0)` in k_wgrad_bf, eight serialised memory latencies per 32 pairs.)
__device__ __forceinline__ float4 load4v(const float* __restrict__ base, int row, int width, int col) {
  const bool ok = row >= 0 && col < width;
  const float4 v = *reinterpret_cast<const float4*>(base + (ok ? (long long)row * width + col : 0ll));
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

struct WgradArgs {
  const float* x;        // [n_in, cin]
  const float* g;        // [n_out, cout]
  const int* hdr;        // map header (null: identity, K = 1)
  const int* nbr;
  const int* rows;
  float* partial;        // [nslices][K][cin][cout]
  long long n_out;
  int cin, cout, K, nslices;
};

// total (segment, slot) work items are enumerated on the host side as K slots of kernel offsets: slot kid in [0,K).
// For transposed / class maps a kernel offset appears in exactly one segment; the kernel finds it in the header.
//
// Round 3 (the training step of BASELINE configs[3] spent 27 of its 43 ms of kernel time here, profiles/r03_train_step_*):
//   * a slice's positions are COMPACTED before they are multiplied: 1024 positions at a time, the present pairs (input row,
//     output row) are gathered into an LDS list in position order and consumed 32 at a time.  A 5x5x5 map on a surface fills
//     30 % of its (offset, position) slots and a 3x3x3 one about half; the first version staged and multiplied the empty
//     ones too.  Skipped terms are exact zeros, so the sums are unchanged bit for bit;
//   * NARROW (cin, cout <= 32: the occupancy heads' 32 -> 16 and 16 -> 1 convolutions over the largest sets, the colour head):
//     the 128 x 128 tile had one 32 x 32 MFMA block of real work in sixteen and three idle waves in four.  Here the four
//     waves split the 32 staged pairs between them, each with one accumulator block, and their blocks are summed in wave
//     order at the end (8.7 ms -> ~1 ms for the two largest launches of the step);
//   * sub-blocks of the 128 x 128 tile that lie outside cin x cout are skipped (wave-uniform).
static constexpr int WG_SUB = 1024;         // positions compacted at a time

template <bool NARROW>
__global__ void __launch_bounds__(256) k_wgrad(WgradArgs a) {
  constexpr int CH = NARROW ? 32 : 128;       // channels staged per operand
  constexpr int LD = CH + 4;
  __shared__ __attribute__((aligned(16))) float Xs[WG_PAIRS * LD];
  __shared__ __attribute__((aligned(16))) float Gs[WG_PAIRS * LD];
  __shared__ int c_in[WG_SUB + WG_PAIRS], c_out[WG_SUB + WG_PAIRS];
  __shared__ int s_wtot[4];
  __shared__ __attribute__((aligned(16))) float red[NARROW ? 4 * 1024 : 4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kid = blockIdx.y;                       // kernel offset id
  const int slice = blockIdx.x;
  const int mt = blockIdx.z / ((a.cout + 127) / 128), nt = blockIdx.z % ((a.cout + 127) / 128);
  const int m0 = mt * 128, n0 = nt * 128;

  // locate the segment / slot that lists this kernel offset
  long long pos_begin = 0, pos_count = a.n_out;
  const int* seg_nbr = nullptr;
  const bool identity = (a.hdr == nullptr);
  if (!identity) {
    const int nseg = a.hdr[HDR_NSEG];
    bool found = false;
    for (int s = 0; s < nseg && !found; ++s) {
      const int* sg = a.hdr + HDR_SEG0 + s * SEG_WORDS;
      for (int j = 0; j < sg[SEG_K_COUNT]; ++j)
        if (a.hdr[HDR_KOFFS + sg[SEG_KOFF_BEGIN] + j] == kid) {
          pos_begin = sg[SEG_POS_BEGIN]; pos_count = sg[SEG_POS_COUNT];
          seg_nbr = a.nbr + (((long long)(unsigned)sg[SEG_NBR_LO]) | ((long long)sg[SEG_NBR_HI] << 32)) + (long long)j * pos_count;
          found = true;
          break;
        }
    }
    if (!found) pos_count = 0;
  }
  const long long per = (pos_count + a.nslices - 1) / a.nslices;
  const long long p_lo = (long long)slice * per;
  const long long p_hi = min(pos_count, p_lo + per);

  const bool vec = ((a.cin | a.cout) & 3) == 0;
  const int wm = w >> 1, wn = w & 1;                // full tile: 2x2 waves, each 64x64 of the 128x128 tile
  const int half = lane >> 5, r31 = lane & 31;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  bool sub_on[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) sub_on[i][j] = m0 + (wm * 2 + i) * 32 < a.cin && n0 + (wn * 2 + j) * 32 < a.cout;

  // one step: the pairs list[0 .. cnt) (cnt <= 32) staged and multiplied
  auto step = [&](const int* l_in, const int* l_out, int cnt) {
    __syncthreads();                                 // previous step's fragment reads are done; the list is complete
    {
      const int r = tid >> 3, part = tid & 7;
      const int ir = r < cnt ? l_in[r] : -1, orow = r < cnt ? l_out[r] : -1;
#pragma unroll
      for (int q = 0; q < (NARROW ? 1 : 4); ++q) {
        const int c = (part + 8 * q) * 4;
        float4 xv, gv;
        if (vec) {                                   // wave-uniform: rows of 4-float multiples, branch-free loads
          xv = load4v(a.x, ir, a.cin, m0 + c);
          gv = load4v(a.g, ir >= 0 ? orow : -1, a.cout, n0 + c);
        } else {
          xv = make_float4(0.f, 0.f, 0.f, 0.f); gv = xv;
          if (ir >= 0) {
            xv = load4(a.x + (long long)ir * a.cin, a.cin, m0 + c);
            gv = load4(a.g + (long long)orow * a.cout, a.cout, n0 + c);
          }
        }
        *reinterpret_cast<float4*>(&Xs[r * LD + c]) = xv;
        *reinterpret_cast<float4*>(&Gs[r * LD + c]) = gv;
      }
    }
    __syncthreads();
    // D[ci][co] += sum_pairs X[pair][ci] * G[pair][co]:  A[i=ci][k=pair], B[k=pair][j=co]
    if (NARROW) {
#pragma unroll
      for (int kk = 0; kk < 8; kk += 2) {            // this wave's quarter of the staged pairs
        const int k0 = w * 8 + kk + half;
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[k0 * LD + r31], Gs[k0 * LD + r31], acc[0][0], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int kk = 0; kk < WG_PAIRS; kk += 2) {
        float af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = Xs[(kk + half) * LD + (wm * 2 + i) * 32 + r31];
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = Gs[(kk + half) * LD + (wn * 2 + j) * 32 + r31];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (sub_on[i][j]) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  int pending = 0;                                   // compacted pairs waiting in c_in / c_out [0, pending)
  for (long long pb = p_lo; pb < p_hi; pb += WG_SUB) {
    // ---- compact the present pairs of positions [pb, pb + 1024): thread t owns positions pb + 4t .. 4t+3 (position order) ----
    int ir[4], cnt = 0;
    const long long q0 = pb + 4 * tid;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long p = q0 + j;
      ir[j] = -1;
      if (p < p_hi) ir[j] = identity ? (int)p : seg_nbr[p];
      cnt += ir[j] >= 0;
    }
    int inc = cnt;                                   // inclusive scan over the wave, then over the four waves
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(inc, d);
      if (lane >= d) inc += t;
    }
    __syncthreads();                                 // (the previous sub-block's steps are done with the list)
    if (lane == 63) s_wtot[w] = inc;
    __syncthreads();
    int off = pending + inc - cnt;
    int total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (i < w) off += s_wtot[i]; total += s_wtot[i]; }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ir[j] >= 0) {
        const long long p = q0 + j;
        c_in[off] = ir[j];
        c_out[off] = a.rows ? a.rows[pos_begin + p] : (int)(pos_begin + p);
        ++off;
      }
    pending += total;
    int g = 0;
    for (; pending - g >= WG_PAIRS; g += WG_PAIRS) step(c_in + g, c_out + g, WG_PAIRS);
    // the remainder (< 32 pairs) moves to the front of the list
    const int rem = pending - g;
    __syncthreads();
    int ti = 0, to = 0;
    if (tid < rem) { ti = c_in[g + tid]; to = c_out[g + tid]; }
    __syncthreads();
    if (tid < rem) { c_in[tid] = ti; c_out[tid] = to; }
    pending = rem;
  }
  if (pending > 0) step(c_in, c_out, pending);

  // partial tile -> [slice][kid][cin][cout]
  float* dst = a.partial + ((long long)slice * a.K + kid) * a.cin * a.cout;
  if (NARROW) {
    // the four waves' blocks, summed in wave order: red[w][e][lane]
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(w * 16 + e) * 64 + lane] = acc[0][0][e];
    __syncthreads();
    if (w == 0) {
      const int co = n0 + r31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = ((red[(0 * 16 + e) * 64 + lane] + red[(1 * 16 + e) * 64 + lane]) + red[(2 * 16 + e) * 64 + lane]) + red[(3 * 16 + e) * 64 + lane];
        const int ci = m0 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (ci < a.cin && co < a.cout) dst[(long long)ci * a.cout + co] = v;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int co = n0 + (wn * 2 + j) * 32 + r31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = m0 + (wm * 2 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (ci < a.cin && co < a.cout) dst[(long long)ci * a.cout + co] = acc[i][j][e];
      }
    }
}

// ------------------------------------------------------------------------------------------
// The same weight gradient on the 16-bit matrix pipe at fp32 accuracy (round 4): both operands split exactly into three bf16
// planes (x = h + m + l, the split of k_conv_mfma_bf), six cross terms per product on v_mfma_f32_32x32x16_bf16 -- 2.67x the
// rate of the fp32-input MFMA the kernel above runs on.  The reduction index of this GEMM is the PAIR, so an MFMA operand needs,
// per channel, 8 consecutive pairs: the staging pass writes the LDS images TRANSPOSED, [plane][channel][pair], a thread packing
// two pairs of one channel into one 32-bit store.  Channel rows are 88 bytes (32 pairs x 2 B + pad): the 32 lanes of a half wave
// (16 pair-pairs x 2 channel groups 8 channels = 176 dwords apart) store to 32 different banks, and the fragment reads -- two
// 8-byte reads per operand, rows 22 dwords apart -- are conflict-free as well.  Tile, slices, compaction of the pair list and the
// fixed summation order are those of k_wgrad<false>.
// ------------------------------------------------------------------------------------------
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wg_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const wg_f32x2 v = {x0, x1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, wg_bf16x2));
  const wg_f32x2 r1 = {x0 - __builtin_bit_cast(float, h << 16), x1 - __builtin_bit_cast(float, h & 0xFFFF0000u)};
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, wg_bf16x2));
  const wg_f32x2 r2 = {r1.x - __builtin_bit_cast(float, m << 16), r1.y - __builtin_bit_cast(float, m & 0xFFFF0000u)};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, wg_bf16x2));
}

static constexpr int WB_ROW = 88;                    // bytes per channel row of a transposed image
static constexpr int WB_PLANE = 128 * WB_ROW;        // one plane: 128 channels

__global__ void __launch_bounds__(256, 2) k_wgrad_bf(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char Xt[3 * WB_PLANE];
  __shared__ __attribute__((aligned(16))) unsigned char Gt[3 * WB_PLANE];
  __shared__ int c_in[WG_SUB + WG_PAIRS], c_out[WG_SUB + WG_PAIRS];
  __shared__ int s_wtot[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kid = blockIdx.y;
  const int slice = blockIdx.x;
  const int mt = blockIdx.z / ((a.cout + 127) / 128), nt = blockIdx.z % ((a.cout + 127) / 128);
  const int m0 = mt * 128, n0 = nt * 128;

  long long pos_begin = 0, pos_count = a.n_out;
  const int* seg_nbr = nullptr;
  const bool identity = (a.hdr == nullptr);
  if (!identity) {
    const int nseg = a.hdr[HDR_NSEG];
    bool found = false;
    for (int s = 0; s < nseg && !found; ++s) {
      const int* sg = a.hdr + HDR_SEG0 + s * SEG_WORDS;
      for (int j = 0; j < sg[SEG_K_COUNT]; ++j)
        if (a.hdr[HDR_KOFFS + sg[SEG_KOFF_BEGIN] + j] == kid) {
          pos_begin = sg[SEG_POS_BEGIN]; pos_count = sg[SEG_POS_COUNT];
          seg_nbr = a.nbr + (((long long)(unsigned)sg[SEG_NBR_LO]) | ((long long)sg[SEG_NBR_HI] << 32)) + (long long)j * pos_count;
          found = true;
          break;
        }
    }
    if (!found) pos_count = 0;
  }
  const long long per = (pos_count + a.nslices - 1) / a.nslices;
  const long long p_lo = (long long)slice * per;
  const long long p_hi = min(pos_count, p_lo + per);

  // wave -> 32 x 32 blocks of the 128 x 128 tile: 2 x 2 waves of 2 x 2 blocks; a tile that is at most 64 wide in one direction is
  // cut the other way (4 x 1 waves of 1 x 2 blocks, or 1 x 4 of 2 x 1; 2 x 2 of 1 x 1 when both are), so that all four waves
  // multiply (128 -> 64: two of the four waves did all the MFMAs)
  const bool thin_m = a.cin - m0 <= 64, thin_n = a.cout - n0 <= 64;
  int bm[2], bn[2];
  bool vm[2], vn[2];
  if (!thin_m && !thin_n) { bm[0] = (w >> 1) * 2; bm[1] = bm[0] + 1; bn[0] = (w & 1) * 2; bn[1] = bn[0] + 1; vm[1] = vn[1] = true; }
  else if (!thin_m) { bm[0] = w; bm[1] = w; bn[0] = 0; bn[1] = 1; vm[1] = false; vn[1] = true; }
  else if (!thin_n) { bm[0] = 0; bm[1] = 1; bn[0] = w; bn[1] = w; vm[1] = true; vn[1] = false; }
  else { bm[0] = bm[1] = w >> 1; bn[0] = bn[1] = w & 1; vm[1] = vn[1] = false; }
  vm[0] = vn[0] = true;
  const int half = lane >> 5, r31 = lane & 31;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  bool sub_on[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) sub_on[i][j] = vm[i] && vn[j] && m0 + bm[i] * 32 < a.cin && n0 + bn[j] * 32 < a.cout;

  // A step = 32 pairs: `issue` starts the global loads of a step's rows into registers, `commit` splits them and writes the
  // transposed LDS images, `compute` multiplies.  Inside a compacted sub-block the loads of step n + 1 are issued before step n
  // is multiplied (the first version waited for every step's rows -- a full memory latency per 32 pairs -- and the 16-bit
  // MFMAs bought nothing: 209 us per launch against 199 for the fp32-input kernel).
  const int pp = tid & 15, cg = tid >> 4;            // pairs 2 pp, 2 pp + 1 of the staged 32; channels cg * 8 .. + 7
  struct Stage { float4 x0[2], x1[2], g0[2], g1[2]; };
  auto issue = [&](Stage& st, const int* l_in, const int* l_out, int cnt) {
    const int r0 = 2 * pp, r1 = r0 + 1;
    const int i0 = r0 < cnt ? l_in[r0] : -1, i1 = r1 < cnt ? l_in[r1] : -1;
    const int o0 = r0 < cnt ? l_out[r0] : -1, o1 = r1 < cnt ? l_out[r1] : -1;
#pragma unroll
    for (int q = 0; q < 2; ++q) {                    // (cin, cout multiples of 4: the dispatcher sends other shapes to k_wgrad)
      const int c = cg * 8 + 4 * q;
      st.x0[q] = load4v(a.x, i0, a.cin, m0 + c);
      st.x1[q] = load4v(a.x, i1, a.cin, m0 + c);
      st.g0[q] = load4v(a.g, i0 >= 0 ? o0 : -1, a.cout, n0 + c);
      st.g1[q] = load4v(a.g, i1 >= 0 ? o1 : -1, a.cout, n0 + c);
    }
  };
  auto commit = [&](const Stage& st) {
    __syncthreads();                                 // previous step's fragment reads are done
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = cg * 8 + 4 * q;
      const float xa[4] = {st.x0[q].x, st.x0[q].y, st.x0[q].z, st.x0[q].w}, xb[4] = {st.x1[q].x, st.x1[q].y, st.x1[q].z, st.x1[q].w};
      const float ga[4] = {st.g0[q].x, st.g0[q].y, st.g0[q].z, st.g0[q].w}, gb[4] = {st.g1[q].x, st.g1[q].y, st.g1[q].z, st.g1[q].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned h, m, l;
        const int off = (c + j) * WB_ROW + pp * 4;
        wg_split2(xa[j], xb[j], h, m, l);
        *reinterpret_cast<unsigned*>(Xt + off) = h;
        *reinterpret_cast<unsigned*>(Xt + WB_PLANE + off) = m;
        *reinterpret_cast<unsigned*>(Xt + 2 * WB_PLANE + off) = l;
        wg_split2(ga[j], gb[j], h, m, l);
        *reinterpret_cast<unsigned*>(Gt + off) = h;
        *reinterpret_cast<unsigned*>(Gt + WB_PLANE + off) = m;
        *reinterpret_cast<unsigned*>(Gt + 2 * WB_PLANE + off) = l;
      }
    }
    __syncthreads();
  };
  // D[ci][co] += sum_pairs X[pair][ci] * G[pair][co]:  A[i = ci][k = pair], B[k = pair][j = co]; 16 pairs per MFMA
  auto compute = [&]() {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      wg_bf16x8 af[3][2], bf[3][2];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned char* src = Xt + p * WB_PLANE + (bm[i] * 32 + r31) * WB_ROW + kb * 32 + half * 16;
          const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
          af[p][i] = __builtin_bit_cast(wg_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const unsigned char* src = Gt + p * WB_PLANE + (bn[j] * 32 + r31) * WB_ROW + kb * 32 + half * 16;
          const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
          bf[p][j] = __builtin_bit_cast(wg_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (sub_on[i][j]) {                          // smallest terms first (planes: 0 = h, 1 = m, 2 = l)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
          }
    }
  };
  Stage sa, sb;
  auto step = [&](const int* l_in, const int* l_out, int cnt) {       // (un-pipelined: the tail of the list)
    __syncthreads();                                                    // the list is complete
    issue(sa, l_in, l_out, cnt);
    commit(sa);
    compute();
  };

  int pending = 0;
  for (long long pb = p_lo; pb < p_hi; pb += WG_SUB) {
    int ir[4], cnt = 0;
    const long long q0 = pb + 4 * tid;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long p = q0 + j;
      ir[j] = -1;
      if (p < p_hi) ir[j] = identity ? (int)p : seg_nbr[p];
      cnt += ir[j] >= 0;
    }
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(inc, d);
      if (lane >= d) inc += t;
    }
    __syncthreads();
    if (lane == 63) s_wtot[w] = inc;
    __syncthreads();
    int off = pending + inc - cnt;
    int total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (i < w) off += s_wtot[i]; total += s_wtot[i]; }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ir[j] >= 0) {
        const long long p = q0 + j;
        c_in[off] = ir[j];
        c_out[off] = a.rows ? a.rows[pos_begin + p] : (int)(pos_begin + p);
        ++off;
      }
    pending += total;
    int g = 0;
    if (pending >= WG_PAIRS) {
      __syncthreads();                               // the list is complete
      // two register stages: the rows of steps n + 1 AND n + 2 are in flight while step n is multiplied (with one stage the
      // loads were outstanding during a third of a step only and the kernel gathered at 1.2 TB/s)
      const int full = pending / WG_PAIRS;           // steps of this sub-block
      issue(sa, c_in, c_out, WG_PAIRS);
      if (full > 1) issue(sb, c_in + WG_PAIRS, c_out + WG_PAIRS, WG_PAIRS);
      for (int st = 0; st < full; st += 2) {
        commit(sa);                                  // rows of step st -> LDS
        if (st + 2 < full) issue(sa, c_in + (st + 2) * WG_PAIRS, c_out + (st + 2) * WG_PAIRS, WG_PAIRS);
        compute();
        if (st + 1 < full) {
          commit(sb);
          if (st + 3 < full) issue(sb, c_in + (st + 3) * WG_PAIRS, c_out + (st + 3) * WG_PAIRS, WG_PAIRS);
          compute();
        }
      }
      g = full * WG_PAIRS;
    }
    const int rem = pending - g;
    __syncthreads();
    int ti = 0, to = 0;
    if (tid < rem) { ti = c_in[g + tid]; to = c_out[g + tid]; }
    __syncthreads();
    if (tid < rem) { c_in[tid] = ti; c_out[tid] = to; }
    pending = rem;
  }
  if (pending > 0) step(c_in, c_out, pending);

  float* dst = a.partial + ((long long)slice * a.K + kid) * a.cin * a.cout;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (!sub_on[i][j]) continue;
      const int co = n0 + bn[j] * 32 + r31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = m0 + bm[i] * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (ci < a.cin && co < a.cout) dst[(long long)ci * a.cout + co] = acc[i][j][e];
      }
    }
}

__global__ void k_wgrad_reduce_wide(const float* __restrict__ partial, long long elems, int nslices, float* __restrict__ dW);
__global__ void k_wgrad_reduce(const float* __restrict__ partial, long long elems, int nslices, float* __restrict__ dW) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= elems) return;
  float s = 0.f;
  for (int sl = 0; sl < nslices; ++sl) s += partial[(long long)sl * elems + t];     // fixed order
  dW[t] = s;
}

// position slices per (offset, tile): enough workgroups to fill the chip (a K = 1 reduction over 20 k rows -- the GDN's gamma
// gradient -- ran on 6 workgroups with 4096-position slices: 0.54 ms for 0.7 GFLOP), at least 64 positions per slice, at most 64
static int wgrad_slices(int64_t n_out, int K, int cin, int cout) {
  const int64_t tiles = (int64_t)((cin + 127) / 128) * ((cout + 127) / 128);
  int64_t s = pcc_cdiv(n_out, WG_SLICE);
  const int64_t fill = pcc_cdiv(512, (int64_t)K * tiles);
  if (s < fill) s = fill;
  const int64_t cap = pcc_cdiv(n_out, 64);
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}

extern "C" size_t pcc_conv_wgrad_ws_bytes(int64_t n_out, int32_t K, int32_t cin, int32_t cout) {
  return (size_t)wgrad_slices(n_out, K, cin, cout) * (size_t)K * cin * cout * sizeof(float) + 256;
}

extern "C" int pcc_conv_wgrad(const float* feat_in, int64_t n_in, int32_t cin, const float* grad_out, int64_t n_out,
                              int32_t cout, int32_t K, const int32_t* hdr, const int32_t* nbr, const int32_t* rows,
                              float* dW, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(dW && K >= 1 && K <= 192 && cin >= 1 && cout >= 1, "pcc_conv_wgrad: bad arguments");
  const long long elems = (long long)K * cin * cout;
  if (n_out <= 0 || n_in <= 0) {
    PCC_CHECK_HIP(hipMemsetAsync(dW, 0, (size_t)elems * sizeof(float), s));
    return PCC_OK;
  }
  PCC_REQUIRE(feat_in && grad_out && ws, "pcc_conv_wgrad: NULL array");
  PCC_REQUIRE(hdr ? (nbr != nullptr) : (K == 1 && n_in == n_out), "pcc_conv_wgrad: map missing (only K=1 may omit it)");
  PCC_REQUIRE(n_in < (1ll << 31) && n_out < (1ll << 31), "pcc_conv_wgrad: too many rows");
  if (ws_bytes < pcc_conv_wgrad_ws_bytes(n_out, K, cin, cout)) {
    pcc_set_error("pcc_conv_wgrad: workspace too small");
    return PCC_EWS;
  }
  WgradArgs a;
  a.x = feat_in; a.g = grad_out; a.hdr = hdr; a.nbr = nbr; a.rows = rows; a.partial = (float*)ws; a.n_out = n_out;
  a.cin = cin; a.cout = cout; a.K = K; a.nslices = wgrad_slices(n_out, K, cin, cout);
  const unsigned tiles = (unsigned)(((cin + 127) / 128) * ((cout + 127) / 128));
  // wide layers: six bf16 terms on the 16-bit pipe (k_wgrad_bf); env PCC_WGRAD_BF=0 keeps the fp32-input MFMA kernel
  static const bool bf_on = getenv("PCC_WGRAD_BF") ? atoi(getenv("PCC_WGRAD_BF")) != 0 : true;
  if (cin <= 32 && cout <= 32) k_wgrad<true><<<dim3((unsigned)a.nslices, (unsigned)K, tiles), 256, 0, s>>>(a);
  else if (bf_on && ((cin | cout) & 3) == 0) k_wgrad_bf<<<dim3((unsigned)a.nslices, (unsigned)K, tiles), 256, 0, s>>>(a);
  else k_wgrad<false><<<dim3((unsigned)a.nslices, (unsigned)K, tiles), 256, 0, s>>>(a);
  PCC_LAUNCH_CHECK();
  if (a.nslices >= 16) k_wgrad_reduce_wide<<<(unsigned)pcc_cdiv(elems, 16), 256, 0, s>>>(a.partial, elems, a.nslices, dW);
  else k_wgrad_reduce<<<(unsigned)pcc_cdiv(elems, 256), 256, 0, s>>>(a.partial, elems, a.nslices, dW);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// Weight gradient of a ONE-logit convolution over a set mapped onto ITSELF (the occupancy heads' second convolution,
// reference `model/transforms.py:146-161` predict_*[2]: cin -> 1, 3x3x3, stride 1 -- in the training step of configs[3] these run over the
// largest candidate sets: 0.98 + 0.91 ms of the step through the pair-list GEMM, whose 128 x 128 / 32 x 32 tiles hold one
// useful column).  For an odd kernel on one set the pair (i, o) of offset k is the pair (o, i) of offset K-1-k, so
//     dW[k][ci] = sum_i  x[i][ci] * g[ nbr_{K-1-k}(i) ]
// is INPUT-stationary: the feature rows stream through once, coalesced, the map is read once, and what is gathered is one
// float per pair from an array that fits the L2.  A thread owns 4 channels of every (256 / LPR)-th row of its block's chunk
// and keeps the K partial sums in registers; lanes are summed by an xor butterfly, waves in wave order, blocks by
// k_wgrad_reduce in block order: deterministic.
// ------------------------------------------------------------------------------------------
static constexpr int WT_KMAX = 27;
static constexpr int WT_MAX_BLOCKS = 1024;
struct WgradThinArgs {
  const float* x;        // [n, cin]
  const float* g;        // [n]
  const int* hdr;
  const int* nbr;
  float* partial;        // [nblocks][K][cin]
  long long n;
  int cin, K, nblocks;
};

template <int LPR>       // lanes per row = cin / 4
__global__ void __launch_bounds__(256, 2) k_wgrad_thin(WgradThinArgs a) {
  constexpr int RPI = 256 / LPR;                    // rows per accumulation sub-step of a workgroup
  constexpr int CIN = LPR * 4;
  constexpr int GS = 29;                            // words per row of the exchange buffer (odd: rows fall on different banks)
  constexpr int RED = 4 * WT_KMAX * CIN;
  __shared__ __attribute__((aligned(16))) int4 s_seg[WT_KMAX];     // {lo, hi of (nbr offset - pos_begin), pos_begin, pos_count}
  __shared__ __attribute__((aligned(16))) float buf[(256 * GS > RED) ? 256 * GS : RED];   // g values [row][k]; at the end: red[w][k][ci]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < WT_KMAX) {                               // segment that lists the INVERSE offset of k
    long long base = 0; int pb = 0, pc = 0;
    const int kk = a.K - 1 - tid;
    if (tid < a.K) {
      const int nseg = a.hdr[HDR_NSEG];
      bool found = false;
      for (int s = 0; s < nseg && !found; ++s) {
        const int* sg = a.hdr + HDR_SEG0 + s * SEG_WORDS;
        for (int j = 0; j < sg[SEG_K_COUNT]; ++j)
          if (a.hdr[HDR_KOFFS + sg[SEG_KOFF_BEGIN] + j] == kk) {
            pb = sg[SEG_POS_BEGIN]; pc = sg[SEG_POS_COUNT];
            base = (((long long)(unsigned)sg[SEG_NBR_LO]) | ((long long)sg[SEG_NBR_HI] << 32)) + (long long)j * pc - pb;
            found = true;
            break;
          }
      }
    }
    s_seg[tid] = make_int4((int)(unsigned)(base & 0xFFFFFFFFll), (int)(base >> 32), pb, pc);
  }
  __syncthreads();
  const int rg = tid / LPR, l = tid % LPR;
  long long per = (a.n + a.nblocks - 1) / a.nblocks;
  per = (per + 255) / 256 * 256;
  const long long lo = (long long)blockIdx.x * per;
  const long long hi = min(a.n, lo + per);
  float4 acc[WT_KMAX];
#pragma unroll
  for (int k = 0; k < WT_KMAX; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  // 256 rows at a time.  Phase 1, thread = row: the row's K inverse neighbours and their gradient values -- every load instruction
  // of a wave covers 64 consecutive rows (the first version had a thread per (row, 4 channels): 4 / 16 lanes asked for the same
  // word and a wave's 54 map / gradient loads served 16 / 4 rows: 0.44 ms for the 992 k-row head).  The values go through LDS
  // to phase 2, thread = (row, 4 channels), which streams the feature rows and accumulates.
  for (long long i0 = lo; i0 < hi; i0 += 256) {
    {
      const long long i = i0 + tid;
      const int4* seg = s_seg;
      asm volatile("" : "+v"(seg));                 // the segment records stay in LDS (hoisted they cost 108 SGPRs and spill)
      int j[WT_KMAX];
#pragma unroll
      for (int k = 0; k < WT_KMAX; ++k) {
        const int4 sg = seg[k];
        const long long off = (((long long)(unsigned)sg.x) | ((long long)sg.y << 32)) + i;
        const unsigned q = (unsigned)(i - sg.z);
        j[k] = (i < hi && k < a.K && q < (unsigned)sg.w) ? a.nbr[off] : -1;
      }
#pragma unroll
      for (int k = 0; k < WT_KMAX; ++k) buf[tid * GS + k] = j[k] >= 0 ? a.g[j[k]] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int sub = 0; sub < LPR; ++sub) {
      const int r = sub * RPI + rg;
      const long long i = i0 + r;
      if (i < hi) {
        const float4 xv = *reinterpret_cast<const float4*>(a.x + i * CIN + 4 * l);
#pragma unroll
        for (int k = 0; k < WT_KMAX; ++k) {
          const float gv = buf[r * GS + k];
          acc[k].x = fmaf(xv.x, gv, acc[k].x); acc[k].y = fmaf(xv.y, gv, acc[k].y);
          acc[k].z = fmaf(xv.z, gv, acc[k].z); acc[k].w = fmaf(xv.w, gv, acc[k].w);
        }
      }
    }
    __syncthreads();
  }
  // lanes holding the same channels: xor butterfly over the row index inside the wave (fixed tree)
  float* red = buf;
#pragma unroll
  for (int k = 0; k < WT_KMAX; ++k) {
#pragma unroll
    for (int d = LPR; d < 64; d <<= 1) {
      acc[k].x += __shfl_xor(acc[k].x, d, 64); acc[k].y += __shfl_xor(acc[k].y, d, 64);
      acc[k].z += __shfl_xor(acc[k].z, d, 64); acc[k].w += __shfl_xor(acc[k].w, d, 64);
    }
    if (lane < LPR) *reinterpret_cast<float4*>(&red[(w * WT_KMAX + k) * CIN + 4 * lane]) = acc[k];
  }
  __syncthreads();
  float* dst = a.partial + (long long)blockIdx.x * a.K * CIN;
  for (int e = tid; e < a.K * CIN; e += 256) {
    const int k = e / CIN, c = e - k * CIN;
    dst[e] = ((red[(0 * WT_KMAX + k) * CIN + c] + red[(1 * WT_KMAX + k) * CIN + c]) + red[(2 * WT_KMAX + k) * CIN + c]) +
             red[(3 * WT_KMAX + k) * CIN + c];
  }
}

// sum of many partial blocks: 16 elements x 16 slice lanes per workgroup, every lane adds its slices in ascending order, the 16
// lane sums are added in lane order (k_wgrad_reduce walks ALL slices in one thread: 0.35 ms for 1024 blocks of 432 values)
__global__ void __launch_bounds__(256) k_wgrad_reduce_wide(const float* __restrict__ partial, long long elems, int nslices,
                                                           float* __restrict__ dW) {
  __shared__ float s[16][17];
  const int el = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const long long e = (long long)blockIdx.x * 16 + el;
  float v = 0.f;
  if (e < elems)
    for (int sl = sg; sl < nslices; sl += 16) v += partial[(long long)sl * elems + e];
  s[sg][el] = v;
  __syncthreads();
  if (sg == 0 && e < elems) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += s[q][el];
    dW[e] = t;
  }
}

// ------------------------------------------------------------------------------------------
// The same input-stationary form for a 16-column gradient (the heads' FIRST convolution over the largest candidate set,
// 32 -> 16 on ~1 M rows: 0.94 ms per training step through the pair-list kernel, which gathers BOTH rows of its 13 M pairs):
//     dW[k][ci][co] = sum_i x[i][ci] * g[nbr_{K-1-k}(i)][co]
// on v_mfma_f32_16x16x4_f32 with both operands straight from global memory: A[ci][row] = one float of a feature row per lane
// (rows streamed in order), B[row][co] = one float of the 64-byte gradient row of the row's inverse neighbour.  A wave owns the
// offsets w, w + 4, ... (<= 7) of a 64-row group: their inverse-neighbour indices are read row-per-lane (one coalesced load
// per offset) and handed to the (row, column) lanes of each 4-row MFMA step by a cross-lane read.  No LDS, no barrier.
// ------------------------------------------------------------------------------------------
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
static constexpr int WS16_KPW = 7;                  // offsets per wave
static constexpr int WS16_MAX_BLOCKS = 1024;

template <int CIN>
__global__ void __launch_bounds__(256, 2) k_wgrad_self16(WgradThinArgs a) {
  constexpr int NB = CIN / 16;
  __shared__ __attribute__((aligned(16))) int4 s_seg[WT_KMAX + 1];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid <= WT_KMAX) {
    long long base = 0; int pb = 0, pc = 0;
    const int kk = a.K - 1 - tid;
    if (tid < a.K) {
      const int nseg = a.hdr[HDR_NSEG];
      bool found = false;
      for (int s = 0; s < nseg && !found; ++s) {
        const int* sg = a.hdr + HDR_SEG0 + s * SEG_WORDS;
        for (int j = 0; j < sg[SEG_K_COUNT]; ++j)
          if (a.hdr[HDR_KOFFS + sg[SEG_KOFF_BEGIN] + j] == kk) {
            pb = sg[SEG_POS_BEGIN]; pc = sg[SEG_POS_COUNT];
            base = (((long long)(unsigned)sg[SEG_NBR_LO]) | ((long long)sg[SEG_NBR_HI] << 32)) + (long long)j * pc - pb;
            found = true;
            break;
          }
      }
    }
    s_seg[tid] = make_int4((int)(unsigned)(base & 0xFFFFFFFFll), (int)(base >> 32), pb, pc);
  }
  __syncthreads();
  long long per = (a.n + a.nblocks - 1) / a.nblocks;
  per = (per + 63) / 64 * 64;
  const long long lo = (long long)blockIdx.x * per;
  const long long hi = min(a.n, lo + per);
  const int c16 = lane & 15, rq = lane >> 4;
  // this wave's offsets and their segment records (wave-uniform: scalar registers)
  long long sbase[WS16_KPW];
  int spb[WS16_KPW], spc[WS16_KPW];
#pragma unroll
  for (int u = 0; u < WS16_KPW; ++u) {
    const int k = w + 4 * u;
    const int4 sg = s_seg[k < a.K ? k : WT_KMAX];      // (record WT_KMAX: empty range)
    const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane(sg.x), bhi = (unsigned)__builtin_amdgcn_readfirstlane(sg.y);
    sbase[u] = ((long long)blo) | ((long long)(int)bhi << 32);
    spb[u] = __builtin_amdgcn_readfirstlane(sg.z);
    spc[u] = k < a.K ? __builtin_amdgcn_readfirstlane(sg.w) : 0;
  }
  wg_f32x4 acc[WS16_KPW][NB];
#pragma unroll
  for (int u = 0; u < WS16_KPW; ++u)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[u][b] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int DEPTH = 4;                              // 4-row MFMA steps whose gradient rows are in flight
  for (long long r0 = lo; r0 < hi; r0 += 64) {
    const long long i = r0 + lane;                      // row-per-lane: the inverse neighbours of this wave's offsets
    int jv[WS16_KPW];
#pragma unroll
    for (int u = 0; u < WS16_KPW; ++u) {
      const unsigned q = (unsigned)(i - spb[u]);
      const bool ok = i < hi && q < (unsigned)spc[u];
      const int v = a.nbr[ok ? sbase[u] + i : 0];
      jv[u] = ok ? v : -1;
    }
    // the 16 x NB feature operands of the group: all loads issued up front (rows beyond the chunk read row `lo`, zeroed at use)
    float araw[16][NB];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const long long row = r0 + 4 * s + rq;
#pragma unroll
      for (int b = 0; b < NB; ++b) araw[s][b] = a.x[(row < hi ? row : lo) * CIN + 16 * b + c16];
    }
    // gradient rows: a ring of DEPTH steps in flight (raw values + a presence mask; the zero of an absent pair is selected at
    // use, so nothing waits on a load before its step is multiplied -- the first version waited per step: 1.09 ms, slower than
    // the pair-list kernel it replaces)
    float graw[DEPTH][WS16_KPW];
    unsigned gmask[DEPTH];
    auto issue = [&](int s, float (&gr)[WS16_KPW], unsigned& m) {
      m = 0;
#pragma unroll
      for (int u = 0; u < WS16_KPW; ++u) {
        const int j = __shfl(jv[u], 4 * s + rq, 64);
        gr[u] = a.g[(long long)(j >= 0 ? j : 0) * 16 + c16];
        m |= (j >= 0 ? 1u : 0u) << u;
      }
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) issue(s, graw[s], gmask[s]);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const bool rok = r0 + 4 * s + rq < hi;
      float av[NB], gv[WS16_KPW];
#pragma unroll
      for (int b = 0; b < NB; ++b) av[b] = rok ? araw[s][b] : 0.f;
#pragma unroll
      for (int u = 0; u < WS16_KPW; ++u) gv[u] = ((gmask[s % DEPTH] >> u) & 1u) ? graw[s % DEPTH][u] : 0.f;
      if (s + DEPTH < 16) issue(s + DEPTH, graw[s % DEPTH], gmask[s % DEPTH]);
#pragma unroll
      for (int u = 0; u < WS16_KPW; ++u)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[u][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[b], gv[u], acc[u][b], 0, 0, 0);
    }
  }
  // D layout: column (co) = lane & 15, row (ci) = 4 * (lane >> 4) + e
  float* dst = a.partial + (long long)blockIdx.x * a.K * CIN * 16;
#pragma unroll
  for (int u = 0; u < WS16_KPW; ++u) {
    const int k = w + 4 * u;
    if (k >= a.K) continue;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[((long long)k * CIN + 16 * b + 4 * rq + e) * 16 + c16] = acc[u][b][e];
  }
}

static bool wgrad_thin_ok(int K, int cin) { return K >= 1 && K <= WT_KMAX && (K & 1) && (cin == 16 || cin == 32 || cin == 64); }
static int wgrad_thin_blocks(int64_t n, int cin) {
  int64_t b = pcc_cdiv(n, 512);                  // at least two 256-row rounds per workgroup
  if (b < 1) b = 1;
  if (b > WT_MAX_BLOCKS) b = WT_MAX_BLOCKS;
  return (int)b;
}
static int wgrad_self16_blocks(int64_t n) {
  int64_t b = pcc_cdiv(n, 512);
  if (b < 1) b = 1;
  if (b > WS16_MAX_BLOCKS) b = WS16_MAX_BLOCKS;
  return (int)b;
}

extern "C" int pcc_conv_wgrad_self_supported(int32_t K, int32_t cin, int32_t cout) {
  return wgrad_thin_ok(K, cin) && (cout == 1 || (cout == 16 && cin <= 32));
}

extern "C" size_t pcc_conv_wgrad_self_ws_bytes(int64_t n, int32_t K, int32_t cin, int32_t cout) {
  const size_t blocks = cout == 1 ? (size_t)wgrad_thin_blocks(n, cin) : (size_t)wgrad_self16_blocks(n);
  return blocks * (size_t)K * cin * (size_t)cout * sizeof(float) + 256;
}

extern "C" int pcc_conv_wgrad_self(const float* feat, int64_t n, int32_t cin, const float* grad_out, int32_t cout, int32_t K,
                                   const int32_t* hdr, const int32_t* nbr, float* dW, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  PCC_REQUIRE(dW && pcc_conv_wgrad_self_supported(K, cin, cout),
              "pcc_conv_wgrad_self: odd K <= 27; cout 1 with cin in {16, 32, 64} or cout 16 with cin in {16, 32} (pcc_conv_wgrad_self_supported)");
  const long long elems = (long long)K * cin * cout;
  if (n <= 0) {
    PCC_CHECK_HIP(hipMemsetAsync(dW, 0, (size_t)elems * sizeof(float), s));
    return PCC_OK;
  }
  PCC_REQUIRE(feat && grad_out && hdr && nbr && ws, "pcc_conv_wgrad_self: NULL array");
  PCC_REQUIRE(n < (1ll << 31), "pcc_conv_wgrad_self: too many rows");
  if (ws_bytes < pcc_conv_wgrad_self_ws_bytes(n, K, cin, cout)) {
    pcc_set_error("pcc_conv_wgrad_self: workspace too small");
    return PCC_EWS;
  }
  WgradThinArgs a;
  a.x = feat; a.g = grad_out; a.hdr = hdr; a.nbr = nbr; a.partial = (float*)ws; a.n = n; a.cin = cin; a.K = K;
  if (cout == 1) {
    a.nblocks = wgrad_thin_blocks(n, cin);
    if (cin == 16) k_wgrad_thin<4><<<(unsigned)a.nblocks, 256, 0, s>>>(a);
    else if (cin == 32) k_wgrad_thin<8><<<(unsigned)a.nblocks, 256, 0, s>>>(a);
    else k_wgrad_thin<16><<<(unsigned)a.nblocks, 256, 0, s>>>(a);
  } else {
    a.nblocks = wgrad_self16_blocks(n);
    if (cin == 16) k_wgrad_self16<16><<<(unsigned)a.nblocks, 256, 0, s>>>(a);
    else k_wgrad_self16<32><<<(unsigned)a.nblocks, 256, 0, s>>>(a);
  }
  PCC_LAUNCH_CHECK();
  k_wgrad_reduce_wide<<<(unsigned)pcc_cdiv(elems, 16), 256, 0, s>>>(a.partial, elems, a.nblocks, dW);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// generative transposed conv backward: dT[pair] = grad_out[output row of the pair]   (every pair written once)
template <typename VT>
__global__ void k_convt_scatter(const VT* __restrict__ g, const int* __restrict__ first, const int* __restrict__ pair_ids,
                                long long n_out, int vpr, VT* __restrict__ dT) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long o = t / vpr;
  if (o >= n_out) return;
  const int v = (int)(t - o * vpr);
  const VT val = g[t];
  for (int q = first[o]; q < first[o + 1]; ++q) dT[(long long)pair_ids[q] * vpr + v] = val;
}

extern "C" int pcc_convt_scatter_rows(const float* grad_out, const int32_t* first, const int32_t* pair_ids, int64_t n_out,
                                      int32_t cout, float* dT, void* stream) {
  if (n_out <= 0) return PCC_OK;
  PCC_REQUIRE(grad_out && first && pair_ids && dT && cout >= 1, "pcc_convt_scatter_rows: bad arguments");
  if (cout % 4 == 0) {
    const int vpr = cout / 4;
    k_convt_scatter<float4><<<(unsigned)pcc_cdiv(n_out * vpr, 256), 256, 0, (hipStream_t)stream>>>(
        (const float4*)grad_out, first, pair_ids, n_out, vpr, (float4*)dT);
  } else {
    k_convt_scatter<float><<<(unsigned)pcc_cdiv(n_out * cout, 256), 256, 0, (hipStream_t)stream>>>(grad_out, first, pair_ids,
                                                                                                 n_out, cout, dT);
  }
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// GDN backward, element-wise parts (reference `model/blocks.py:38-57`, GDN1 form; the matrix products run on the convolution
// kernels: n = beta + |x| gamma^T, v = u gamma, d gamma = u^T |x|).  The torch chain was ~40 launches per GDN layer, most of
// them on [C] / [C, C] parameter tensors (the NonNegativeParametrizer and its gradient through autograd.grad).
//   pre :  GDN  y = x / n : u = -g x / n^2, dx0 = g / n        IGDN  y = x n : u = g x, dx0 = g n
//   post:  dx = dx0 + sign(x) v
//   gamma_eff = max(gamma_raw, bound)^2 - 2^-36 (CompressAI's reparametrisation, as pcc_gdn_pack applies it)
//   reparam_bwd: d raw = [raw >= bound or d eff < 0] * 2 max(raw, bound) * d eff   (LowerBound's gradient rule)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gdn_bwd_pre(const float4* __restrict__ x, const float4* __restrict__ g,
                                                     const float4* __restrict__ n, long long total4, int inverse,
                                                     float4* __restrict__ u, float4* __restrict__ dx0) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total4) return;
  const float4 xv = x[t], gv = g[t], nv = n[t];
  float4 uo, d0;
  if (inverse) {
    uo = make_float4(gv.x * xv.x, gv.y * xv.y, gv.z * xv.z, gv.w * xv.w);
    d0 = make_float4(gv.x * nv.x, gv.y * nv.y, gv.z * nv.z, gv.w * nv.w);
  } else {
    uo = make_float4(-(gv.x * xv.x) / (nv.x * nv.x), -(gv.y * xv.y) / (nv.y * nv.y), -(gv.z * xv.z) / (nv.z * nv.z),
                     -(gv.w * xv.w) / (nv.w * nv.w));
    d0 = make_float4(gv.x / nv.x, gv.y / nv.y, gv.z / nv.z, gv.w / nv.w);
  }
  u[t] = uo;
  dx0[t] = d0;
}

__device__ __forceinline__ float sgn1(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ void __launch_bounds__(256) k_gdn_bwd_post(float4* __restrict__ dx, const float4* __restrict__ x,
                                                      const float4* __restrict__ v, long long total4) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total4) return;
  const float4 xv = x[t], vv = v[t];
  float4 d = dx[t];
  d.x += sgn1(xv.x) * vv.x; d.y += sgn1(xv.y) * vv.y; d.z += sgn1(xv.z) * vv.z; d.w += sgn1(xv.w) * vv.w;
  dx[t] = d;
}

__global__ void k_gdn_gamma_eff(const float* __restrict__ gamma_raw, int cc, float bound, float pedestal, float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cc) return;
  const float gq = fmaxf(gamma_raw[t], bound);
  out[t] = gq * gq - pedestal;
}

// d_gamma_t: [ci][co] as pcc_conv_wgrad(|x|, u) returns it; gamma_raw / d_gamma_raw: [co][ci]
__global__ void k_gdn_reparam_bwd(const float* __restrict__ beta_raw, const float* __restrict__ gamma_raw,
                                  const float* __restrict__ d_beta, const float* __restrict__ d_gamma_t, int c, float beta_bound,
                                  float gamma_bound, float* __restrict__ d_beta_raw, float* __restrict__ d_gamma_raw) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < c) {
    const float r = beta_raw[t], d = d_beta[t];
    d_beta_raw[t] = (r >= beta_bound || d < 0.f) ? 2.f * fmaxf(r, beta_bound) * d : 0.f;
  }
  if (t >= c * c) return;
  const int co = t / c, ci = t - co * c;
  const float r = gamma_raw[t], d = d_gamma_t[ci * c + co];
  d_gamma_raw[t] = (r >= gamma_bound || d < 0.f) ? 2.f * fmaxf(r, gamma_bound) * d : 0.f;
}

extern "C" int pcc_gdn_bwd_pre(const float* x, const float* g, const float* n, int64_t elems, int32_t inverse, float* u, float* dx0,
                               void* stream) {
  PCC_REQUIRE(elems >= 0 && elems % 4 == 0 && (elems == 0 || (x && g && n && u && dx0)), "pcc_gdn_bwd_pre: bad arguments");
  if (elems == 0) return PCC_OK;
  k_gdn_bwd_pre<<<(unsigned)pcc_cdiv(elems / 4, 256), 256, 0, (hipStream_t)stream>>>((const float4*)x, (const float4*)g, (const float4*)n,
                                                                                   elems / 4, inverse, (float4*)u, (float4*)dx0);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_gdn_bwd_post(float* dx, const float* x, const float* v, int64_t elems, void* stream) {
  PCC_REQUIRE(elems >= 0 && elems % 4 == 0 && (elems == 0 || (dx && x && v)), "pcc_gdn_bwd_post: bad arguments");
  if (elems == 0) return PCC_OK;
  k_gdn_bwd_post<<<(unsigned)pcc_cdiv(elems / 4, 256), 256, 0, (hipStream_t)stream>>>((float4*)dx, (const float4*)x, (const float4*)v, elems / 4);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_gdn_gamma_eff(const float* gamma_raw, int32_t c, float* gamma_eff, void* stream) {
  PCC_REQUIRE(gamma_raw && gamma_eff && c >= 1, "pcc_gdn_gamma_eff: bad arguments");
  const double pedestal = 1.0 / 68719476736.0;   // 2^-36, as pcc_gdn_pack
  k_gdn_gamma_eff<<<(unsigned)pcc_cdiv((int64_t)c * c, 256), 256, 0, (hipStream_t)stream>>>(gamma_raw, c * c, (float)sqrt(pedestal),
                                                                                          (float)pedestal, gamma_eff);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_gdn_reparam_bwd(const float* beta_raw, const float* gamma_raw, const float* d_beta, const float* d_gamma_t,
                                   int32_t c, float beta_min, float* d_beta_raw, float* d_gamma_raw, void* stream) {
  PCC_REQUIRE(beta_raw && gamma_raw && d_beta && d_gamma_t && d_beta_raw && d_gamma_raw && c >= 1, "pcc_gdn_reparam_bwd: bad arguments");
  const double pedestal = 1.0 / 68719476736.0;
  k_gdn_reparam_bwd<<<(unsigned)pcc_cdiv((int64_t)c * c, 256), 256, 0, (hipStream_t)stream>>>(
      beta_raw, gamma_raw, d_beta, d_gamma_t, c, (float)sqrt((double)beta_min + pedestal), (float)sqrt(pedestal), d_beta_raw,
      d_gamma_raw);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}
