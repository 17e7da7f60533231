// Hyperprior likelihood path: fused Gaussian-conditional kernel (scale bound, table index search,
// quantisation, likelihood) and the factorised prior on the hyper latent.  Element-wise, HBM-bound.
#include "pcc_common.h"

static constexpr float SCALE_BOUND = 0.11f;
static constexpr float LIK_BOUND = 1e-9f;

__device__ inline float std_cum(float x) {           // 0.5 * erfc(-x / sqrt(2))
  return 0.5f * erfcf(-0.70710678118654752440f * x);
}

__device__ inline int table_index(float s, const float* __restrict__ tab, int nt) {
  // idx = (nt-1) - #{t < nt-1 : s <= tab[t]}   (GaussianConditional.build_indexes)
  int idx = nt - 1;
  for (int t = 0; t < nt - 1; ++t) idx -= (s <= tab[t]) ? 1 : 0;
  return idx;
}

template <bool ENCODE>
__global__ void __launch_bounds__(256) k_gauss(const float* __restrict__ y, const int* __restrict__ sym_in,
                                               const float* __restrict__ params, const int64_t* __restrict__ keys,
                                               const float* __restrict__ gain, long long n, int c,
                                               const float* __restrict__ table, int nt, int* __restrict__ sym,
                                               int* __restrict__ idx, float* __restrict__ lik,
                                               float* __restrict__ y_hat) {
  __shared__ float tab[256];
  if (threadIdx.x < nt) tab[threadIdx.x] = table[threadIdx.x];
  __syncthreads();
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * c) return;
  const long long r = t / c;
  const int ch = (int)(t - r * c);
  const float scale = params[r * 2 * c + ch];
  const float mean = params[r * 2 * c + c + ch];
  float g = 1.f;
  if (gain) g = gain[(keys[r] >> 48) * c + ch];
  const float s = fmaxf(scale * g, SCALE_BOUND);
  if (idx) idx[t] = table_index(s, tab, nt);
  if (ENCODE) {
    if (!sym && !lik) return;                       // index-only call (decoder side, before the symbols exist)
    const float q = rintf(y[t] * g - mean * g);
    if (sym) sym[t] = (int)q;
    if (lik) {
      const float a = fabsf(q);
      const float up = std_cum((0.5f - a) / s);
      const float lo = std_cum((-0.5f - a) / s);
      lik[t] = fmaxf(up - lo, LIK_BOUND);
    }
  } else {
    y_hat[t] = (float)sym_in[t] + mean * g;
  }
}

extern "C" int pcc_gauss_encode(const float* y, const float* params, const int64_t* keys, const float* gain,
                                int64_t n, int32_t c, const float* table, int32_t n_table, int32_t* sym, int32_t* idx,
                                float* lik, void* stream) {
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(params && table && c >= 1 && n_table >= 2 && n_table <= 256, "pcc_gauss_encode: bad arguments");
  PCC_REQUIRE(y || (!sym && !lik), "pcc_gauss_encode: y is NULL (allowed only for an index-only call)");
  PCC_REQUIRE(!gain || keys, "pcc_gauss_encode: gain needs keys");
  k_gauss<true><<<(unsigned)pcc_cdiv(n * c, 256), 256, 0, (hipStream_t)stream>>>(y, nullptr, params, keys, gain, n, c,
                                                                                 table, n_table, sym, idx, lik, nullptr);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_gauss_decode(const int32_t* sym, const float* params, const int64_t* keys, const float* gain,
                                int64_t n, int32_t c, const float* table, int32_t n_table, float* y_hat, int32_t* idx,
                                void* stream) {
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(sym && params && table && y_hat && c >= 1 && n_table >= 2 && n_table <= 256,
              "pcc_gauss_decode: bad arguments");
  PCC_REQUIRE(!gain || keys, "pcc_gauss_decode: gain needs keys");
  k_gauss<false><<<(unsigned)pcc_cdiv(n * c, 256), 256, 0, (hipStream_t)stream>>>(nullptr, sym, params, keys, gain, n,
                                                                                  c, table, n_table, nullptr, idx,
                                                                                  nullptr, y_hat);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// factorised prior, filters (3,3,3,3): packed [c][58]
// ------------------------------------------------------------------------------------------
__device__ inline float eb_logits(const float* __restrict__ p, float x) {
  const float* m0 = p;        const float* M1 = p + 3;  const float* M2 = p + 12; const float* M3 = p + 21;
  const float* m4 = p + 30;   const float* b = p + 33;  const float* f = p + 46;
  float h[3], g[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { h[i] = m0[i] * x + b[i]; h[i] += f[i] * tanhf(h[i]); }
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const float* M = (l == 0) ? M1 : (l == 1 ? M2 : M3);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = M[o * 3 + 0] * h[0];
      v += M[o * 3 + 1] * h[1];
      v += M[o * 3 + 2] * h[2];
      v += b[3 + 3 * l + o];
      g[o] = v + f[3 + 3 * l + o] * tanhf(v);
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) h[o] = g[o];
  }
  float v = m4[0] * h[0];
  v += m4[1] * h[1];
  v += m4[2] * h[2];
  return v + b[12];
}

__device__ inline float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void __launch_bounds__(256) k_eb_encode(const float* __restrict__ z, long long n, int c,
                                                   const float* __restrict__ packed, const float* __restrict__ med,
                                                   int* __restrict__ sym, float* __restrict__ z_hat,
                                                   float* __restrict__ lik) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * c) return;
  const int ch = (int)(t % c);
  const float m = med[ch];
  const float q = rintf(z[t] - m);
  const float zh = q + m;
  if (sym) sym[t] = (int)q;
  if (z_hat) z_hat[t] = zh;
  if (lik) {
    const float* p = packed + (long long)ch * 58;
    const float lo = eb_logits(p, zh - 0.5f);
    const float up = eb_logits(p, zh + 0.5f);
    const float sum = lo + up;
    const float sg = (sum > 0.f) ? -1.f : ((sum < 0.f) ? 1.f : 0.f);
    lik[t] = fmaxf(fabsf(sigmoidf_(sg * up) - sigmoidf_(sg * lo)), LIK_BOUND);
  }
}

extern "C" int pcc_eb_encode(const float* z, int64_t n, int32_t c, const float* eb_packed, const float* medians,
                             int32_t* sym, float* z_hat, float* lik, void* stream) {
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(z && medians && c >= 1 && (!lik || eb_packed), "pcc_eb_encode: bad arguments");
  k_eb_encode<<<(unsigned)pcc_cdiv(n * c, 256), 256, 0, (hipStream_t)stream>>>(z, n, c, eb_packed, medians, sym, z_hat,
                                                                             lik);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// Differentiable likelihood of the factorised prior (training forward: `EntropyBottleneck._likelihood` + lower bound 1e-9;
// reference `model/entropy_models.py:272,282-285`, rate term `loss.py:77-79`).  The torch chain is ~35 element-wise / batched
// matmul launches per direction and per evaluation point (lower, upper); here one kernel per direction.
//   lik = max(|sigmoid(sg up) - sigmoid(sg lo)|, 1e-9),  lo / up = logits(v -+ 0.5),  sg = -sign(lo + up) (no gradient)
// backward: d v per element, and the gradient of the 58 PACKED parameters of every channel (softplus(matrices) | biases |
// tanh(factors): the reparametrisations are differentiated by torch on the [C, 58] tensor), summed over the rows by a
// workgroup per channel in a fixed order (deterministic).  LowerBound rule of CompressAI on the bound.
// ------------------------------------------------------------------------------------------
struct EbTape { float a[4][3]; float h[5][3]; };       // pre-activations of the four gated layers, inputs of the five linear maps

__device__ inline float eb_logits_tape(const float* __restrict__ p, float x, EbTape& t) {
  const float* m0 = p;        const float* Ms = p + 3;  const float* m4 = p + 30;   const float* b = p + 33;  const float* f = p + 46;
#pragma unroll
  for (int i = 0; i < 3; ++i) { t.a[0][i] = m0[i] * x + b[i]; t.h[1][i] = t.a[0][i] + f[i] * tanhf(t.a[0][i]); }
#pragma unroll
  for (int l = 1; l < 4; ++l) {
    const float* M = Ms + 9 * (l - 1);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = M[o * 3 + 0] * t.h[l][0];
      v += M[o * 3 + 1] * t.h[l][1];
      v += M[o * 3 + 2] * t.h[l][2];
      v += b[3 * l + o];
      t.a[l][o] = v;
      t.h[l + 1][o] = v + f[3 * l + o] * tanhf(v);
    }
  }
  float v = m4[0] * t.h[4][0];
  v += m4[1] * t.h[4][1];
  v += m4[2] * t.h[4][2];
  return v + b[12];
}

// reverse pass of eb_logits for the upstream gradient gout: dp[58] += d out / d packed * gout; returns d out / d x * gout
__device__ inline float eb_logits_bwd(const float* __restrict__ p, float x, const EbTape& t, float gout, float* __restrict__ dp) {
  const float* m0 = p;        const float* Ms = p + 3;  const float* m4 = p + 30;   const float* f = p + 46;
  float dh[3];
  dp[33 + 12] += gout;
#pragma unroll
  for (int j = 0; j < 3; ++j) { dp[30 + j] += gout * t.h[4][j]; dh[j] = gout * m4[j]; }
#pragma unroll
  for (int l = 3; l >= 1; --l) {
    const float* M = Ms + 9 * (l - 1);
    float da[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const float th = tanhf(t.a[l][o]);
      dp[46 + 3 * l + o] += dh[o] * th;
      da[o] = dh[o] * (1.f + f[3 * l + o] * (1.f - th * th));
      dp[33 + 3 * l + o] += da[o];
#pragma unroll
      for (int j = 0; j < 3; ++j) dp[3 + 9 * (l - 1) + o * 3 + j] += da[o] * t.h[l][j];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) dh[j] = da[0] * M[0 * 3 + j] + da[1] * M[1 * 3 + j] + da[2] * M[2 * 3 + j];
  }
  float dx = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float th = tanhf(t.a[0][i]);
    dp[46 + i] += dh[i] * th;
    const float da = dh[i] * (1.f + f[i] * (1.f - th * th));
    dp[33 + i] += da;
    dp[i] += da * x;
    dx += da * m0[i];
  }
  return dx;
}

__global__ void __launch_bounds__(256) k_eb_lik_fwd(const float* __restrict__ v, long long n, int c, const float* __restrict__ packed,
                                                    float* __restrict__ lik) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * c) return;
  const float* p = packed + (long long)(t % c) * 58;
  const float lo = eb_logits(p, v[t] - 0.5f), up = eb_logits(p, v[t] + 0.5f);
  const float sum = lo + up;
  const float sg = (sum > 0.f) ? -1.f : ((sum < 0.f) ? 1.f : 0.f);
  lik[t] = fmaxf(fabsf(sigmoidf_(sg * up) - sigmoidf_(sg * lo)), LIK_BOUND);
}

// one workgroup per channel: rows strided over the threads, 58 partial sums per thread, fixed-order reduction
__global__ void __launch_bounds__(256) k_eb_lik_bwd(const float* __restrict__ v, const float* __restrict__ g, long long n, int c,
                                                    const float* __restrict__ packed, float* __restrict__ dv,
                                                    float* __restrict__ dpacked) {
  __shared__ float red[4][58];
  const int ch = blockIdx.x;
  const float* p = packed + (long long)ch * 58;
  float dp[58];
#pragma unroll
  for (int i = 0; i < 58; ++i) dp[i] = 0.f;
  for (long long r = threadIdx.x; r < n; r += 256) {
    const long long t = r * c + ch;
    const float x = v[t];
    EbTape tl, tu;
    const float lo = eb_logits_tape(p, x - 0.5f, tl), up = eb_logits_tape(p, x + 0.5f, tu);
    const float sum = lo + up;
    const float sg = (sum > 0.f) ? -1.f : ((sum < 0.f) ? 1.f : 0.f);
    const float su = sigmoidf_(sg * up), sl = sigmoidf_(sg * lo);
    const float d = su - sl;
    float gl = g[t];
    if (!(fabsf(d) >= LIK_BOUND || gl < 0.f)) gl = 0.f;               // LowerBound(1e-9) on the likelihood
    const float sd = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
    const float gup = gl * sd * su * (1.f - su) * sg, glo = -gl * sd * sl * (1.f - sl) * sg;
    float dx = eb_logits_bwd(p, x + 0.5f, tu, gup, dp);
    dx += eb_logits_bwd(p, x - 0.5f, tl, glo, dp);
    if (dv) dv[t] = dx;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 58; ++i) {
    float s = dp[i];
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1) s += __shfl_xor(s, dlt);
    if (lane == 0) red[w][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 58) dpacked[(long long)ch * 58 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

extern "C" int pcc_eb_lik_fwd(const float* v, int64_t n, int32_t c, const float* eb_packed, float* lik, void* stream) {
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(v && eb_packed && lik && c >= 1, "pcc_eb_lik_fwd: bad arguments");
  k_eb_lik_fwd<<<(unsigned)pcc_cdiv(n * c, 256), 256, 0, (hipStream_t)stream>>>(v, n, c, eb_packed, lik);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_eb_lik_bwd(const float* v, const float* grad_lik, int64_t n, int32_t c, const float* eb_packed, float* dv,
                              float* d_packed, void* stream) {
  PCC_REQUIRE(eb_packed && d_packed && c >= 1 && n >= 0 && (n == 0 || (v && grad_lik)), "pcc_eb_lik_bwd: bad arguments");
  k_eb_lik_bwd<<<(unsigned)c, 256, 0, (hipStream_t)stream>>>(v, grad_lik, n, c, eb_packed, dv, d_packed);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// Differentiable Gaussian likelihood of the training forward (`GaussianConditional._likelihood` +
// likelihood lower bound; reference call sites `model/entropy_models.py:312-316,327-331`, rate term `loss.py:77-79`):
//   s = max(scale, 0.11), a = |v - mean|,  lik = max(Phi((.5 - a)/s) - Phi((-.5 - a)/s), 1e-9)
// forward and backward as one element-wise kernel each (the torch chain is ~15 launches per direction).  Gradients
// follow CompressAI's LowerBound rule: a bounded quantity passes the gradient when it is inside the bound or when the
// gradient would move it back inside (g < 0).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gauss_lik_fwd(const float* __restrict__ v, const float* __restrict__ scale,
                                                       const float* __restrict__ mean, long long n, float* __restrict__ lik) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const float s = fmaxf(scale[t], SCALE_BOUND);
  const float a = fabsf(v[t] - (mean ? mean[t] : 0.f));
  lik[t] = fmaxf(std_cum((0.5f - a) / s) - std_cum((-0.5f - a) / s), LIK_BOUND);
}

__global__ void __launch_bounds__(256) k_gauss_lik_bwd(const float* __restrict__ v, const float* __restrict__ scale,
                                                       const float* __restrict__ mean, const float* __restrict__ g,
                                                       long long n, float* __restrict__ dv, float* __restrict__ dscale,
                                                       float* __restrict__ dmean) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const float sc = scale[t];
  const float s = fmaxf(sc, SCALE_BOUND);
  const float d = v[t] - (mean ? mean[t] : 0.f);
  const float a = fabsf(d);
  const float u1 = (0.5f - a) / s, u2 = (-0.5f - a) / s;
  const float raw = std_cum(u1) - std_cum(u2);
  float gl = g[t];
  if (!(raw >= LIK_BOUND || gl < 0.f)) gl = 0.f;                      // LowerBound(1e-9) on the likelihood
  const float inv_sqrt_2pi = 0.39894228040143267794f;
  const float p1 = inv_sqrt_2pi * expf(-0.5f * u1 * u1), p2 = inv_sqrt_2pi * expf(-0.5f * u2 * u2);
  const float dl_da = (p2 - p1) / s;
  const float dl_ds = (u2 * p2 - u1 * p1) / s;
  const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);          // torch.abs: zero gradient at 0
  const float gv = gl * dl_da * sg;
  if (dv) dv[t] = gv;
  if (dmean) dmean[t] = -gv;
  if (dscale) {
    const float gs = gl * dl_ds;
    dscale[t] = (sc >= SCALE_BOUND || gs < 0.f) ? gs : 0.f;          // LowerBound(0.11) on the scale
  }
}

extern "C" int pcc_gauss_lik_fwd(const float* v, const float* scale, const float* mean, int64_t n, float* lik, void* stream) {
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(v && scale && lik, "pcc_gauss_lik_fwd: NULL array");
  k_gauss_lik_fwd<<<(unsigned)pcc_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(v, scale, mean, n, lik);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_gauss_lik_bwd(const float* v, const float* scale, const float* mean, const float* grad_lik, int64_t n,
                                 float* dv, float* dscale, float* dmean, void* stream) {
  if (n <= 0) return PCC_OK;
  PCC_REQUIRE(v && scale && grad_lik, "pcc_gauss_lik_bwd: NULL array");
  k_gauss_lik_bwd<<<(unsigned)pcc_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(v, scale, mean, grad_lik, n, dv, dscale, dmean);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// Quantisation-offset network of the training forward (`quant_nn`, reference `model/entropy_models.py:210-233`, call sites
// `:318-322`): per ELEMENT of the latent a 2 -> 10 -> 10 -> 1 perceptron on (gain, standard deviation).  As torch layers it is
// three GEMMs with 1.6 M rows and 2 / 10 / 10 columns per direction plus the ReLUs; their weight gradients reduce over all
// elements into 10 x 10 outputs, which hipBLASLt runs on ONE 16 x 16 tile (0.54 ms of the training step), and the chain is
// ~25 launches.  Here: one element-wise kernel per direction, the 151 parameters in registers / scalar loads, hidden layers
// recomputed in the backward pass, parameter gradients summed per thread, per wave (xor butterfly), per workgroup and then
// over the workgroups in index order: deterministic.
//   params: W1 [10][2] | b1 [10] | W2 [10][10] | b2 [10] | W3 [10] | b3      (torch.nn.Linear layouts, row = output)
// ------------------------------------------------------------------------------------------
static constexpr int QM_H = 10;
static constexpr int QM_W1 = 0, QM_B1 = 20, QM_W2 = 30, QM_B2 = 130, QM_W3 = 140, QM_B3 = 150, QM_PARAMS = 151;
static constexpr int QM_BLOCKS = 1024;

__device__ __forceinline__ float qm_forward(const float* __restrict__ p, float s, float d, float (&h1)[QM_H], float (&h2)[QM_H]) {
#pragma unroll
  for (int j = 0; j < QM_H; ++j) h1[j] = fmaxf(fmaf(p[QM_W1 + 2 * j], s, fmaf(p[QM_W1 + 2 * j + 1], d, p[QM_B1 + j])), 0.f);
#pragma unroll
  for (int j = 0; j < QM_H; ++j) {
    float t = p[QM_B2 + j];
#pragma unroll
    for (int i = 0; i < QM_H; ++i) t = fmaf(p[QM_W2 + QM_H * j + i], h1[i], t);
    h2[j] = fmaxf(t, 0.f);
  }
  float o = p[QM_B3];
#pragma unroll
  for (int j = 0; j < QM_H; ++j) o = fmaf(p[QM_W3 + j], h2[j], o);
  return o;
}

__global__ void __launch_bounds__(256) k_quant_mlp_fwd(const float* __restrict__ scale, const float* __restrict__ stddev,
                                                       long long n, const float* __restrict__ params, float* __restrict__ out) {
  __shared__ float p[QM_PARAMS];
  if (threadIdx.x < QM_PARAMS) p[threadIdx.x] = params[threadIdx.x];
  __syncthreads();
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float h1[QM_H], h2[QM_H];
  out[t] = qm_forward(p, scale[t], stddev[t], h1, h2);
}

__global__ void __launch_bounds__(256) k_quant_mlp_bwd(const float* __restrict__ scale, const float* __restrict__ stddev,
                                                       const float* __restrict__ go, long long n, const float* __restrict__ params,
                                                       float* __restrict__ d_scale, float* __restrict__ d_stddev,
                                                       float* __restrict__ partial /*[gridDim.x][151]*/) {
  __shared__ float p[QM_PARAMS];
  __shared__ float red[4][QM_PARAMS];
  if (threadIdx.x < QM_PARAMS) p[threadIdx.x] = params[threadIdx.x];
  __syncthreads();
  float acc[QM_PARAMS];
#pragma unroll
  for (int i = 0; i < QM_PARAMS; ++i) acc[i] = 0.f;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
    const float s = scale[t], d = stddev[t], g = go[t];
    float h1[QM_H], h2[QM_H];
    qm_forward(p, s, d, h1, h2);
    acc[QM_B3] += g;
    float dh2[QM_H], dh1[QM_H];
#pragma unroll
    for (int j = 0; j < QM_H; ++j) {
      acc[QM_W3 + j] = fmaf(g, h2[j], acc[QM_W3 + j]);
      dh2[j] = h2[j] > 0.f ? g * p[QM_W3 + j] : 0.f;
      acc[QM_B2 + j] += dh2[j];
    }
#pragma unroll
    for (int i = 0; i < QM_H; ++i) dh1[i] = 0.f;
#pragma unroll
    for (int j = 0; j < QM_H; ++j)
#pragma unroll
      for (int i = 0; i < QM_H; ++i) {
        acc[QM_W2 + QM_H * j + i] = fmaf(dh2[j], h1[i], acc[QM_W2 + QM_H * j + i]);
        dh1[i] = fmaf(p[QM_W2 + QM_H * j + i], dh2[j], dh1[i]);
      }
    float ds = 0.f, dd = 0.f;
#pragma unroll
    for (int i = 0; i < QM_H; ++i) {
      const float e = h1[i] > 0.f ? dh1[i] : 0.f;
      acc[QM_B1 + i] += e;
      acc[QM_W1 + 2 * i] = fmaf(e, s, acc[QM_W1 + 2 * i]);
      acc[QM_W1 + 2 * i + 1] = fmaf(e, d, acc[QM_W1 + 2 * i + 1]);
      ds = fmaf(p[QM_W1 + 2 * i], e, ds);
      dd = fmaf(p[QM_W1 + 2 * i + 1], e, dd);
    }
    if (d_scale) d_scale[t] = ds;
    if (d_stddev) d_stddev[t] = dd;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < QM_PARAMS; ++i) {
    float v = acc[i];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0) red[w][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < QM_PARAMS)
    partial[(long long)blockIdx.x * QM_PARAMS + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// d_params[i] = sum over the workgroups' partial sums, in workgroup order (one workgroup; 151 x <= 1024 values)
__global__ void __launch_bounds__(256) k_quant_mlp_sum(const float* __restrict__ partial, int nblocks, float* __restrict__ d_params) {
  if (threadIdx.x >= QM_PARAMS) return;
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += partial[(long long)b * QM_PARAMS + threadIdx.x];
  d_params[threadIdx.x] = s;
}

static int qm_blocks(int64_t n) {
  int64_t b = pcc_cdiv(n, 1024);                 // >= 4 elements per thread
  return (int)(b < 1 ? 1 : (b > QM_BLOCKS ? QM_BLOCKS : b));
}

extern "C" int32_t pcc_quant_mlp_params(void) { return QM_PARAMS; }
extern "C" size_t pcc_quant_mlp_ws_bytes(int64_t n) { return (size_t)qm_blocks(n) * QM_PARAMS * sizeof(float) + 256; }

extern "C" int pcc_quant_mlp_fwd(const float* scale, const float* stddev, int64_t n, const float* params, float* out, void* stream) {
  PCC_REQUIRE(params && n >= 0 && (n == 0 || (scale && stddev && out)), "pcc_quant_mlp_fwd: bad arguments");
  if (n == 0) return PCC_OK;
  k_quant_mlp_fwd<<<(unsigned)pcc_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(scale, stddev, n, params, out);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

extern "C" int pcc_quant_mlp_bwd(const float* scale, const float* stddev, const float* grad_out, int64_t n, const float* params,
                                 float* d_scale, float* d_stddev, float* d_params, void* ws, size_t ws_bytes, void* stream) {
  PCC_REQUIRE(params && d_params && n >= 0 && (n == 0 || (scale && stddev && grad_out && ws)), "pcc_quant_mlp_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    PCC_CHECK_HIP(hipMemsetAsync(d_params, 0, QM_PARAMS * sizeof(float), s));
    return PCC_OK;
  }
  if (ws_bytes < pcc_quant_mlp_ws_bytes(n)) {
    pcc_set_error("pcc_quant_mlp_bwd: workspace too small");
    return PCC_EWS;
  }
  const int nb = qm_blocks(n);
  k_quant_mlp_bwd<<<(unsigned)nb, 256, 0, s>>>(scale, stddev, grad_out, n, params, d_scale, d_stddev, (float*)ws);
  PCC_LAUNCH_CHECK();
  k_quant_mlp_sum<<<1, 256, 0, s>>>((const float*)ws, nb, d_params);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}

// ------------------------------------------------------------------------------------------
// Focal loss of one occupancy level of the training forward (`Multiscale_FocalLoss`, reference `loss.py:115-157`): per
// candidate row  p = sigmoid(logit),  v = occupied ? p : 1 - p,  pt = clip(v, 1e-2, 1),
//     f = -(occupied ? alpha : 1 - alpha) * (1 - pt)^gamma * log(pt) * w[batch(row)]
// and its derivative with respect to the logit (clip passes the gradient inside its range, as torch.clip does).  The torch
// chain is ~12 launches forward and ~20 backward per level; this is one, the level's mean is a sum of `f` divided by n.
// occ_row: result of pcc_lookup_rows against the ground-truth set (>= 0: occupied).  w = q_map[batch][0] (row pitch q_pitch).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_focal_rows(const float* __restrict__ logits, long long stride,
                                                    const int* __restrict__ occ_row, const long long* __restrict__ keys,
                                                    long long n, const float* __restrict__ q_map, int q_pitch, float alpha,
                                                    float gamma, float* __restrict__ f, float* __restrict__ df) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const bool occ = occ_row[t] >= 0;
  const float x = logits[t * stride];
  const float p = 1.f / (1.f + expf(-x));
  const float v = occ ? p : 1.f - p;
  const bool inside = v >= 1e-2f;                      // (v <= 1 always)
  const float pt = inside ? v : 1e-2f;
  const float a = occ ? alpha : 1.f - alpha;
  const float w = q_map[(keys[t] >> 48) * q_pitch];
  const float om = 1.f - pt;
  const float pw = powf(om, gamma), lg = logf(pt);
  f[t] = -a * pw * lg * w;
  // d f / d pt = -a w ( -gamma om^(gamma-1) log pt + om^gamma / pt )
  const float pw1 = om > 0.f ? powf(om, gamma - 1.f) : (gamma > 1.f ? 0.f : 1.f);
  const float dfdpt = -a * w * (-gamma * pw1 * lg + pw / pt);
  df[t] = inside ? dfdpt * (occ ? 1.f : -1.f) * p * (1.f - p) : 0.f;
}

extern "C" int pcc_focal_rows(const float* logits, int64_t stride_elems, const int32_t* occ_row, const int64_t* keys, int64_t n,
                              const float* q_map, int32_t q_pitch, float alpha, float gamma, float* f, float* df, void* stream) {
  PCC_REQUIRE(n >= 0 && stride_elems >= 1 && q_pitch >= 1 && (n == 0 || (logits && occ_row && keys && q_map && f && df)),
              "pcc_focal_rows: bad arguments");
  if (n == 0) return PCC_OK;
  k_focal_rows<<<(unsigned)pcc_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(logits, stride_elems, occ_row, (const long long*)keys, n,
                                                                           q_map, q_pitch, alpha, gamma, f, df);
  PCC_LAUNCH_CHECK();
  return PCC_OK;
}
