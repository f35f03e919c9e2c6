// fp32-tolerance inference mode (HRIEMO_PRECISION=fp32; include/hriemo.h, "fp32-tolerance mode"): the reference computes in
// fp32 throughout (models/cross_modal_block_tacfn.py:70-125, beta_gate_tacfn.py:68-118, emotion_decoder.py:30-64,116-162); the
// bf16 product path is 1e-2 away from it, this mode 1e-3 or better.  gfx950 has no fast fp32 matrix path for GEMM-sized work
// (157 TFLOP/s fp32 MFMA against 2.5 PFLOP/s bf16), so
//   * Linear layers run on the bf16 GEMM kernel with operands split in three (x = hi + mid + lo, hi = bf16(x), mid = bf16(x - hi)):
//     x.w ~= hi.hi + mid.hi + hi.mid (the dropped terms are <= 2^-16 |x||w|), written as ONE GEMM over a 3K-long contraction:
//     activations become [hi | mid | hi], weights [hi | hi | mid] (split3_kernel), fp32 accumulate and fp32 output;
//   * the attention cores (a few % of the FLOPs) run on the fp32 MFMA itself (v_mfma_f32_16x16x4_f32), S^T = K.Q^T so that the
//     probabilities come out of the accumulator registers already in the B-operand layout of O^T = V^T.P^T;
//   * LayerNorm, pooling, the gate and the fusion are fp32 row kernels.
// Round 4: the backward of every piece in the same arithmetic (the IEMOCAP trainer runs fp32 without autocast,
// scripts/fusion/train_fusion_seq_level_decoder.py:310-334): dX / dW on the same 3K-contraction bf16 GEMM (operands split along
// the contraction -- rows for the transposed layouts, split3_kernel forms 2 / 3), attention backward on the fp32 MFMA, fp32
// LayerNorm / gate / head backward row kernels.  Dropout is not built into these kernels: the mode trains with dropout = 0 (the
// host side refuses p > 0 under autograd instead of silently dropping nothing).
#include "common.h"
#include <math.h>

// ------------------------------------------------------------------------------------------- operand splitting
// Y[m][0:K] = hi, Y[m][K:2K] = layout ? hi : mid, Y[m][2K:3K] = layout ? mid : hi   (layout 0: activations, 1: weights)
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ X, long ldx, int M, int K, bf16_t* __restrict__ Y, int layout,
                                                     int relu) {
  const int kq = K >> 2;
  const long nv = (long)M * kq;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (long)gridDim.x * blockDim.x) {
    const long m = v / kq;
    const int k = (int)(v - m * kq) * 4;
    f32x4 x = *(const f32x4*)(X + m * ldx + k);
    bf16x4 hi, mid;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = x[j];
      if (relu) t = fmaxf(t, 0.f);
      hi[j] = (bf16_t)t;
      mid[j] = (bf16_t)(t - (float)hi[j]);
    }
    bf16_t* y = Y + m * 3L * K + k;
    *(bf16x4*)y = hi;
    *(bf16x4*)(y + K) = layout ? hi : mid;
    *(bf16x4*)(y + 2 * K) = layout ? mid : hi;
  }
}

extern "C" int hriemo_split_bf16x3(const float* X, long ldx, int M, int K, void* Y, int layout, int relu, hipStream_t st) {
  HRIEMO_CHECK(M > 0 && K > 0 && K % 8 == 0 && ldx % 4 == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0, "split_bf16x3: bad shape or alignment");
  const long nv = (long)M * (K >> 2);
  int grid = (int)((nv + 255) / 256);
  if (grid > 8192) grid = 8192;
  hriemo_prof_begin(HP_ROWOPS, st);
  hipLaunchKernelGGL(split3_kernel, dim3(grid), dim3(256), 0, st, X, ldx, M, K, (bf16_t*)Y, layout, relu);
  HRIEMO_LAUNCH_CHECK("split3_kernel");
  hriemo_prof_end(HP_ROWOPS, st, (double)M * K * 10.0);
  return 0;
}

// ------------------------------------------------------------------------------------------- dropout in the fp32 kernels
// The same counter hash, keys and element indices as the bf16 kernels (common.h; rowops.hip: key = site_key(seed, site, 0), a = row
// of the padded layout, b = column; attention.hip: key = site_key(seed, site, (b_offset + b) * H + h), a = query, b = key position),
// so a model draws the same masks in either precision and tests/hashrng.py replays both.  Evaluated element by element (keep16):
// these kernels are not the product path's hot loop.
struct RowDrop {
  uint32_t thr16; float inv_keep; uint64_t seed; const unsigned long long* seed_dev; uint32_t site; long row_off;
};
static RowDrop make_row_drop(float p, uint64_t seed, const unsigned long long* seed_dev, uint32_t site, long row_off) {
  const DropCfg c = make_drop(p, seed, site);
  RowDrop r; r.thr16 = c.thr16; r.inv_keep = c.inv_keep; r.seed = seed; r.seed_dev = seed_dev; r.site = site; r.row_off = row_off;
  return r;
}
struct AttnDrop {
  uint32_t thr16; float inv_keep; uint64_t seed; const unsigned long long* seed_dev; uint32_t site; int b_offset;
};
static AttnDrop make_attn_drop(float p, uint64_t seed, const unsigned long long* seed_dev, uint32_t site, int b_offset) {
  const DropCfg c = make_drop(p, seed, site);
  AttnDrop r; r.thr16 = c.thr16; r.inv_keep = c.inv_keep; r.seed = seed; r.seed_dev = seed_dev; r.site = site; r.b_offset = b_offset;
  return r;
}
__device__ __forceinline__ f32x4 drop4(f32x4 v, uint32_t key, uint32_t a, uint32_t b0, uint32_t thr16, float inv_keep) {
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = keep16(key, a, b0 + (uint32_t)j, thr16) ? v[j] * inv_keep : 0.f;
  return v;
}
// 0 / inv_keep per element: the factor d drop(g) / d g
__device__ __forceinline__ f32x4 drop4_factor(uint32_t key, uint32_t a, uint32_t b0, uint32_t thr16, float inv_keep) {
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = keep16(key, a, b0 + (uint32_t)j, thr16) ? inv_keep : 0.f;
  return v;
}

// ------------------------------------------------------------------------------------------- element-wise dropout (the FFN's hidden layer)
// Y = drop(relu ? max(X, 0) : X), times (gate > 0) when a gate matrix is given.  Forward: X = the hidden pre-activations, relu = 1
// (emotion_decoder.py:57: linear2(dropout(relu(linear1(x))))).  Backward: X = the gradient of the dropped activations, gate = the
// pre-activations (ReLU's derivative), same site and row offset: the same keep / (1 - p) factors.
__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* __restrict__ X, float* __restrict__ Y, long M, int N, int relu,
                                                          const float* __restrict__ gate, RowDrop dr) {
  const int nq = N >> 2;
  const long nv = M * nq;
  const uint32_t dkey = dr.thr16 != 0 ? site_key(eff_seed(dr.seed, dr.seed_dev), dr.site, 0u) : 0u;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (long)gridDim.x * blockDim.x) {
    const long m = v / nq;
    const int c = (int)(v - m * nq) * 4;
    f32x4 x = *(const f32x4*)(X + m * N + c);
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = fmaxf(x[j], 0.f);
    }
    if (gate != nullptr) {
      const f32x4 gt = *(const f32x4*)(gate + m * N + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = gt[j] > 0.f ? x[j] : 0.f;
    }
    if (dr.thr16 != 0) x = drop4(x, dkey, (uint32_t)(m + dr.row_off), (uint32_t)c, dr.thr16, dr.inv_keep);
    *(f32x4*)(Y + m * N + c) = x;
  }
}
extern "C" int hriemo_dropout_f32(const float* X, float* Y, long M, int N, int relu, const float* gate, float p, uint64_t seed,
                                  const unsigned long long* seed_dev, uint32_t site, long row_off, hipStream_t st) {
  HRIEMO_CHECK(M > 0 && N > 0 && N % 4 == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)gate % 16) == 0,
               "dropout_f32: N=%d must be a multiple of 4 and the matrices 16-byte aligned", N);
  HRIEMO_CHECK(p >= 0.f && p < 1.f, "dropout_f32: dropout p=%f", (double)p);
  const RowDrop dr = make_row_drop(p, seed, seed_dev, site, row_off);
  long g = (M * (N >> 2) + 255) / 256;
  if (g > 8192) g = 8192;
  hriemo_prof_begin(HP_ROWOPS, st);
  hipLaunchKernelGGL(dropout_f32_kernel, dim3((int)g), dim3(256), 0, st, X, Y, M, N, relu, gate, dr);
  HRIEMO_LAUNCH_CHECK("dropout_f32_kernel");
  hriemo_prof_end(HP_ROWOPS, st, (double)M * N * (gate != nullptr ? 12.0 : 8.0));
  return 0;
}

// ------------------------------------------------------------------------------------------- LayerNorm(x + drop(g)), fp32 in and out
// one wave per row, the row in registers (d <= 64 * 4 * NV4)
template <int NV4>
__global__ __launch_bounds__(256) void add_ln_f32_kernel(const float* __restrict__ G, const float* __restrict__ X, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ Y32, bf16_t* __restrict__ Y16, int M,
                                                         int d, float eps, RowDrop dr) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nq = d >> 2;
  const uint32_t dkey = dr.thr16 != 0 ? site_key(eff_seed(dr.seed, dr.seed_dev), dr.site, 0u) : 0u;
  f32x4 s[NV4];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NV4; ++c) {
    const int q = lane + 64 * c;
    s[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (q < nq) {
      s[c] = *(const f32x4*)(G + row * d + q * 4);
      if (dr.thr16 != 0) s[c] = drop4(s[c], dkey, (uint32_t)(row + dr.row_off), (uint32_t)(q * 4), dr.thr16, dr.inv_keep);
      if (X != nullptr) s[c] += *(const f32x4*)(X + row * d + q * 4);
      sum += s[c][0] + s[c][1] + s[c][2] + s[c][3];
    }
  }
  const float mu = wave_sum(sum) / (float)d;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < NV4; ++c)
    if (lane + 64 * c < nq) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float t = s[c][j] - mu; sq += t * t; }
    }
  const float rstd = 1.f / sqrtf(wave_sum(sq) / (float)d + eps);
#pragma unroll
  for (int c = 0; c < NV4; ++c) {
    const int q = lane + 64 * c;
    if (q < nq) {
      const f32x4 gm = *(const f32x4*)(gamma + q * 4), bt = *(const f32x4*)(beta + q * 4);
      f32x4 o;
      bf16x4 o16;
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] = (s[c][j] - mu) * rstd * gm[j] + bt[j]; o16[j] = (bf16_t)o[j]; }
      *(f32x4*)(Y32 + row * d + q * 4) = o;
      if (Y16 != nullptr) *(bf16x4*)(Y16 + row * d + q * 4) = o16;
    }
  }
}

extern "C" int hriemo_add_ln_f32(const float* G, const float* X, const float* gamma, const float* beta, float* Y32, void* Y16, int M, int d,
                                 float eps, float p, uint64_t seed, const unsigned long long* seed_dev, uint32_t site, long row_off,
                                 hipStream_t st) {
  HRIEMO_CHECK(M > 0 && d > 0 && d % 4 == 0 && d <= 4096, "add_ln_f32: d=%d must be a multiple of 4, at most 4096", d);
  HRIEMO_CHECK(p >= 0.f && p < 1.f, "add_ln_f32: dropout p=%f", (double)p);
  const RowDrop dr = make_row_drop(p, seed, seed_dev, site, row_off);
  hriemo_prof_begin(HP_ROWOPS, st);
  const dim3 grid((M + 3) / 4);
  if (d <= 1024) hipLaunchKernelGGL((add_ln_f32_kernel<4>), grid, dim3(256), 0, st, G, X, gamma, beta, Y32, (bf16_t*)Y16, M, d, eps, dr);
  else hipLaunchKernelGGL((add_ln_f32_kernel<16>), grid, dim3(256), 0, st, G, X, gamma, beta, Y32, (bf16_t*)Y16, M, d, eps, dr);
  HRIEMO_LAUNCH_CHECK("add_ln_f32_kernel");
  hriemo_prof_end(HP_ROWOPS, st, (double)M * d * 14.0);
  return 0;
}

// ------------------------------------------------------------------------------------------- gate pieces (beta_gate_tacfn.py)
// masked mean over the sequence (:6-24): pooled[b][c] = sum_{valid l} X[b][l][c] / max(#valid, 1); fixed summation order
__global__ __launch_bounds__(256) void masked_mean_f32_kernel(const float* __restrict__ X, const uint8_t* __restrict__ mask, float* __restrict__ pooled,
                                                              int L, int d) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  float s = 0.f, n = 0.f;
  for (int l = 0; l < L; ++l) {
    const bool valid = mask == nullptr || mask[(long)b * L + l] == 0;
    if (valid) { s += X[((long)b * L + l) * d + c]; n += 1.f; }
  }
  pooled[(long)b * d + c] = s / fmaxf(n, 1.f);
}
// gate input [a, t, |a - t|, a * t] (:87-89)
__global__ __launch_bounds__(256) void gate_in_f32_kernel(const float* __restrict__ a, const float* __restrict__ t, float* __restrict__ gin, int d) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const float av = a[(long)b * d + c], tv = t[(long)b * d + c];
  float* g = gin + (long)b * 4 * d;
  g[c] = av; g[d + c] = tv; g[2 * d + c] = fabsf(av - tv); g[3 * d + c] = av * tv;
}
// w = sigmoid(pre) with the accurate expf, beta = mean_d(w) (:92-95)
__global__ __launch_bounds__(256) void sigmoid_beta_f32_kernel(const float* __restrict__ pre, float* __restrict__ w, float* __restrict__ beta, int d) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int c = tid; c < d; c += 256) {
    const float v = 1.f / (1.f + expf(-pre[(long)b * d + c]));
    w[(long)b * d + c] = v;
    s += v;
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) beta[b] = (red[0] + red[1] + red[2] + red[3]) / (float)d;
}
// h[b][l] = w[b] * A[b][l] + (1 - w[b]) * T[b][l] over the first L positions (:98-116); A, T have their own sequence lengths
__global__ __launch_bounds__(256) void fuse_f32_kernel(const float* __restrict__ w, const float* __restrict__ A, int La, const float* __restrict__ T,
                                                       int Lt, float* __restrict__ H32, bf16_t* __restrict__ H16, int B, int L, int d) {
  const int nq = d >> 2;
  const long nv = (long)B * L * nq;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (long)gridDim.x * blockDim.x) {
    const int q = (int)(v % nq);
    const long bl = v / nq;
    const int l = (int)(bl % L);
    const long b = bl / L;
    const f32x4 a = *(const f32x4*)(A + ((long)b * La + l) * d + q * 4), t = *(const f32x4*)(T + ((long)b * Lt + l) * d + q * 4);
    const f32x4 wv = *(const f32x4*)(w + b * d + q * 4);
    f32x4 o;
    bf16x4 o16;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = wv[j] * a[j] + (1.f - wv[j]) * t[j]; o16[j] = (bf16_t)o[j]; }
    *(f32x4*)(H32 + v * 4) = o;
    if (H16 != nullptr) *(bf16x4*)(H16 + v * 4) = o16;
  }
}

extern "C" int hriemo_masked_mean_f32(const float* X, const unsigned char* mask, float* pooled, int B, int L, int d, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && L > 0 && d > 0, "masked_mean_f32: empty input");
  hipLaunchKernelGGL(masked_mean_f32_kernel, dim3((d + 255) / 256, B), dim3(256), 0, st, X, mask, pooled, L, d);
  HRIEMO_LAUNCH_CHECK("masked_mean_f32_kernel");
  return 0;
}
extern "C" int hriemo_gate_input_f32(const float* a_pool, const float* t_pool, float* gate_in, int B, int d, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && d > 0, "gate_input_f32: empty input");
  hipLaunchKernelGGL(gate_in_f32_kernel, dim3((d + 255) / 256, B), dim3(256), 0, st, a_pool, t_pool, gate_in, d);
  HRIEMO_LAUNCH_CHECK("gate_in_f32_kernel");
  return 0;
}
extern "C" int hriemo_sigmoid_beta_f32(const float* pre, float* w, float* beta, int B, int d, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && d > 0, "sigmoid_beta_f32: empty input");
  hipLaunchKernelGGL(sigmoid_beta_f32_kernel, dim3(B), dim3(256), 0, st, pre, w, beta, d);
  HRIEMO_LAUNCH_CHECK("sigmoid_beta_f32_kernel");
  return 0;
}
extern "C" int hriemo_fuse_f32(const float* w, const float* A, int La, const float* T, int Lt, float* H32, void* H16, int B, int L, int d,
                               hipStream_t st) {
  HRIEMO_CHECK(B > 0 && L > 0 && d % 4 == 0 && L <= La && L <= Lt, "fuse_f32: bad shape (L=%d La=%d Lt=%d d=%d)", L, La, Lt, d);
  const long nv = (long)B * L * (d >> 2);
  int grid = (int)((nv + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(fuse_f32_kernel, dim3(grid), dim3(256), 0, st, w, A, La, T, Lt, H32, (bf16_t*)H16, B, L, d);
  HRIEMO_LAUNCH_CHECK("fuse_f32_kernel");
  return 0;
}

// ------------------------------------------------------------------------------------------- attention on the fp32 MFMA
// One wave = 16 queries of one (batch, head); a block = 4 waves = 64 queries; key tiles of 64 staged in LDS as fp32 rows.
// v_mfma_f32_16x16x4_f32: lane l = (i = l & 15, g = l >> 4) supplies A[row i][k g] and B[k g][col i], holds C[4g + r][i].
//   S^T[key][query]  = sum_dim K[key][dim] Q[query][dim]   : A = K (LDS), B = Q^T (registers, HD/4 values per lane)
//   O^T[dim][query] += sum_key V[key][dim] P^T[key][query] : A = V^T (LDS), B = P^T -- lane (i, g) holds P^T[16n + 4g + r][i] in
//   accumulator register r of sub-tile n, so MFMA step (n, r) contracts over the keys {16n + 4g' + r : g' = 0..3} with both
//   operands indexed by g: no transpose, no LDS round trip for P.
template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ Q, long ldq, const float* __restrict__ K, long ldk,
                                                           const float* __restrict__ V, long ldv, float* __restrict__ O, long ldo,
                                                           const uint8_t* __restrict__ kpm, float* __restrict__ lse, int H, int Lq, int Lk,
                                                           float scale, AttnDrop dr) {
  constexpr int LDR = HD + 4;                 // LDS row stride in floats
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Ks = (float*)smem_raw;               // [64][LDR]
  float* Vs = Ks + 64 * LDR;
  float* pad = Vs + 64 * LDR;                 // [64] additive mask of the key tile: 0 or -inf
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int q = min(q0 + i, Lq - 1);
  const float* qp = Q + ((long)b * Lq + q) * ldq + h * HD;
  float qf[HD / 4];
#pragma unroll
  for (int ks = 0; ks < HD / 4; ++ks) qf[ks] = qp[4 * ks + g] * scale;
  f32x4 o[HD / 16];
#pragma unroll
  for (int t = 0; t < HD / 16; ++t) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun = -INFINITY, lrun = 0.f;          // running max (shared by the 4 lanes of a query) and THIS lane's partial sum
  const uint32_t dkey = dr.thr16 != 0 ? site_key(eff_seed(dr.seed, dr.seed_dev), dr.site, (uint32_t)((dr.b_offset + b) * H + h)) : 0u;
  const float* Kb = K + (long)b * Lk * ldk + h * HD;
  const float* Vb = V + (long)b * Lk * ldv + h * HD;
  for (int k0 = 0; k0 < Lk; k0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * (HD / 4); e += 256) {
      const int r = e / (HD / 4), c = (e - r * (HD / 4)) * 4;
      const int key = k0 + r;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (key < Lk) { kv = *(const f32x4*)(Kb + (long)key * ldk + c); vv = *(const f32x4*)(Vb + (long)key * ldv + c); }
      *(f32x4*)(Ks + r * LDR + c) = kv;
      *(f32x4*)(Vs + r * LDR + c) = vv;
    }
    if (tid < 64) {
      const int key = k0 + tid;
      pad[tid] = (key < Lk && (kpm == nullptr || kpm[(long)b * Lk + key] == 0)) ? 0.f : -INFINITY;
    }
    __syncthreads();
    f32x4 s[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      s[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HD / 4; ++ks)
        s[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(16 * n + i) * LDR + 4 * ks + g], qf[ks], s[n], 0, 0, 0);
    }
    float mt = -INFINITY;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[n][r] += pad[16 * n + 4 * g + r];
        mt = fmaxf(mt, s[n][r]);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 16));
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float mnew = fmaxf(mrun, mt);
    const float msafe = mnew == -INFINITY ? 0.f : mnew;           // fully masked so far: every exponent below is exp(-inf) = 0
    const float corr = expf(mrun - msafe);                        // mrun = -inf: 0
    mrun = mnew;
    lrun *= corr;
#pragma unroll
    for (int t = 0; t < HD / 16; ++t) o[t] *= corr;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = expf(s[n][r] - msafe);
        lrun += p;                              // the softmax denominator sums every key; dropout acts on the normalised weights
        s[n][r] = (dr.thr16 == 0 || keep16(dkey, (uint32_t)(q0 + i), (uint32_t)(k0 + 16 * n + 4 * g + r), dr.thr16)) ? p : 0.f;
      }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < HD / 16; ++t)
          o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[(16 * n + 4 * g + r) * LDR + 16 * t + i], s[n][r], o[t], 0, 0, 0);
  }
  float l = lrun + __shfl_xor(lrun, 16);
  l += __shfl_xor(l, 32);
  // a query whose keys are all padding: PyTorch's softmax over -inf gives NaN, and so does this (0 * inf)
  const float inv = (dr.thr16 != 0 ? dr.inv_keep : 1.f) / l;
  if (q0 + i < Lq) {
    float* op = O + ((long)b * Lq + q0 + i) * ldo + h * HD;
#pragma unroll
    for (int t = 0; t < HD / 16; ++t) {
      f32x4 v = o[t];
      v *= inv;
      if (l == 0.f) v = (f32x4){NAN, NAN, NAN, NAN};
      *(f32x4*)(op + 16 * t + 4 * g) = v;
    }
    if (g == 0 && lse != nullptr) lse[((long)b * H + h) * Lq + q0 + i] = l == 0.f ? -INFINITY : mrun + logf(l);
  }
}

// head-averaged probabilities [B, Lq, Lk] (need_weights=True, average_attn_weights=True): block = 64 queries of one batch entry,
// key tiles outer, heads inner; p = exp(s - lse) summed over heads in registers, times 1/H
template <int HD>
__global__ __launch_bounds__(256) void attn_probs_f32_kernel(const float* __restrict__ Q, long ldq, const float* __restrict__ K, long ldk,
                                                             const uint8_t* __restrict__ kpm, const float* __restrict__ lse,
                                                             float* __restrict__ probs, int H, int Lq, int Lk, float scale, AttnDrop dr) {
  constexpr int LDR = HD + 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Ks = (float*)smem_raw;
  float* pad = Ks + 64 * LDR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int q = min(q0 + i, Lq - 1);
  const float invH = 1.f / (float)H;
  for (int k0 = 0; k0 < Lk; k0 += 64) {
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < H; ++h) {
      __syncthreads();
      const float* Kb = K + (long)b * Lk * ldk + h * HD;
      for (int e = tid; e < 64 * (HD / 4); e += 256) {
        const int r = e / (HD / 4), c = (e - r * (HD / 4)) * 4;
        const int key = k0 + r;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f};
        if (key < Lk) kv = *(const f32x4*)(Kb + (long)key * ldk + c);
        *(f32x4*)(Ks + r * LDR + c) = kv;
      }
      if (h == 0 && tid < 64) {
        const int key = k0 + tid;
        pad[tid] = (key < Lk && (kpm == nullptr || kpm[(long)b * Lk + key] == 0)) ? 0.f : -INFINITY;
      }
      __syncthreads();
      const float* qp = Q + ((long)b * Lq + q) * ldq + h * HD;
      const float ls = lse[((long)b * H + h) * Lq + q];
      // (training mode: nn.MultiheadAttention returns the weights AFTER its dropout, so the export drops what the forward dropped)
      const uint32_t dkey = dr.thr16 != 0 ? site_key(eff_seed(dr.seed, dr.seed_dev), dr.site, (uint32_t)((dr.b_offset + b) * H + h)) : 0u;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < HD / 4; ++ks)
          s = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(16 * n + i) * LDR + 4 * ks + g], qp[4 * ks + g] * scale, s, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = expf(s[r] + pad[16 * n + 4 * g + r] - ls);                           // lse = -inf (no valid key): NaN, as PyTorch
          if (dr.thr16 == 0) acc[n][r] += pv;
          else acc[n][r] += keep16(dkey, (uint32_t)(q0 + i), (uint32_t)(k0 + 16 * n + 4 * g + r), dr.thr16) ? pv * dr.inv_keep : 0.f * pv;
        }
      }
    }
    if (q0 + i < Lq) {
      float* pp = probs + ((long)b * Lq + q0 + i) * Lk + k0;
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * n + 4 * g + r;
          if (k0 + key < Lk) pp[key] = acc[n][r] * invH;
        }
    }
  }
}


// ===================================================================================================== backward (round 4)
// ------------------------------------------------------------------------------------------- operand splitting, general form
// form 0: Y[M][3K] = [hi | mid | hi]   form 1: Y[M][3K] = [hi | hi | mid]      (contraction along the columns of X)
// form 2: Y[3M][K] = [hi ; mid ; hi]   form 3: Y[3M][K] = [hi ; hi ; mid]      (contraction along the rows of X: dW = dY^T . X,
//                                                                                 and the weight operand of dX = dY . W)
// form 4: Y[M][6K] = [hi | mid | lo | hi | mid | hi]   form 5: Y[M][6K] = [hi | hi | hi | mid | mid | lo]   (x = hi + mid + lo EXACTLY)
// form 6: Y[6M][K] = [hi ; mid ; lo ; hi ; mid ; hi]   form 7: Y[6M][K] = [hi ; hi ; hi ; mid ; mid ; lo]   (the same along the rows)
// x is first multiplied by (mask > 0) when a mask is given (ReLU's derivative on the FFN's hidden gradient) and clamped at 0
// when relu is set.  Pairing form 0 / 2 with form 1 / 3 yields hi.hi + mid.hi + hi.mid (2^-16 relative: the backward GEMMs).
// Pairing form 4 / 6 with form 5 / 7 yields all six products down to 2^-24 (hi.hi + mid.hi + lo.hi + hi.mid + mid.mid + hi.lo):
// what hri-emo_amd/_fp32.py uses for every GEMM of the mode since round 4.  The forward needs it -- one that is 4e-6 off flips the
// sign of a ReLU pre-activation that lies that close to zero, and one flipped unit among the 40 x 4096 of a small batch moves its
// weight row's gradient by 2e-3 of the whole matrix (measured: the 3-product forward left FFN first-layer gradients 0.6-5e-3
// off, everything smooth at 1e-5) -- and the backward needs it on ill-conditioned weights: the closed-form fixtures amplify a
// GEMM's error ~400x into the gate MLP's gradient (3-product backward 1.7e-3 there, the fp32 reference itself 2e-5).
__global__ __launch_bounds__(256) void split3g_kernel(const float* __restrict__ X, long ldx, int M, int K, bf16_t* __restrict__ Y, int form,
                                                      int relu, const float* __restrict__ mask, long ldm) {
  const int kq = K >> 2;
  const long nv = (long)M * kq;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (long)gridDim.x * blockDim.x) {
    const long m = v / kq;
    const int k = (int)(v - m * kq) * 4;
    f32x4 x = *(const f32x4*)(X + m * ldx + k);
    if (mask != nullptr) {
      const f32x4 mk = *(const f32x4*)(mask + m * ldm + k);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = mk[j] > 0.f ? x[j] : 0.f;
    }
    bf16x4 hi, mid, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = x[j];
      if (relu) t = fmaxf(t, 0.f);
      hi[j] = (bf16_t)t;
      const float r1 = t - (float)hi[j];
      mid[j] = (bf16_t)r1;
      lo[j] = (bf16_t)(r1 - (float)mid[j]);
    }
    const bool wform = form & 1;          // weight-type order: hi, hi, mid
    if (form >= 4) {
      // six blocks, side by side (forms 4 / 5: block stride K inside a row of 6K) or stacked (forms 6 / 7: block stride M * K)
      const long bs = form < 6 ? (long)K : (long)M * K;
      bf16_t* y = form < 6 ? Y + m * 6L * K + k : Y + m * (long)K + k;
      *(bf16x4*)y = hi;
      *(bf16x4*)(y + bs) = wform ? hi : mid;
      *(bf16x4*)(y + 2 * bs) = wform ? hi : lo;
      *(bf16x4*)(y + 3 * bs) = wform ? mid : hi;
      *(bf16x4*)(y + 4 * bs) = mid;
      *(bf16x4*)(y + 5 * bs) = wform ? lo : hi;
    } else if (form < 2) {
      bf16_t* y = Y + m * 3L * K + k;
      *(bf16x4*)y = hi;
      *(bf16x4*)(y + K) = wform ? hi : mid;
      *(bf16x4*)(y + 2 * K) = wform ? mid : hi;
    } else {
      bf16_t* y = Y + m * (long)K + k;
      const long blk = (long)M * K;
      *(bf16x4*)y = hi;
      *(bf16x4*)(y + blk) = wform ? hi : mid;
      *(bf16x4*)(y + 2 * blk) = wform ? mid : hi;
    }
  }
}
extern "C" int hriemo_split3_f32(const float* X, long ldx, int M, int K, void* Y, int form, int relu, const float* mask, long ldmask,
                                 hipStream_t st) {
  HRIEMO_CHECK(M > 0 && K > 0 && K % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 8) == 0 && form >= 0 && form <= 7,
               "split3_f32: bad shape, alignment or form");
  HRIEMO_CHECK(mask == nullptr || (ldmask % 4 == 0 && ((uintptr_t)mask % 16) == 0), "split3_f32: mask alignment");
  const long nv = (long)M * (K >> 2);
  int grid = (int)((nv + 255) / 256);
  if (grid > 8192) grid = 8192;
  hriemo_prof_begin(HP_ROWOPS, st);
  hipLaunchKernelGGL(split3g_kernel, dim3(grid), dim3(256), 0, st, X, ldx, M, K, (bf16_t*)Y, form, relu, mask, ldmask);
  HRIEMO_LAUNCH_CHECK("split3g_kernel");
  hriemo_prof_end(HP_ROWOPS, st, (double)M * K * 10.0);
  return 0;
}

// ------------------------------------------------------------------------------------------- column sums of an fp32 matrix
// out[n] (+)= sum_m X[m][n], fixed order: slice partials [S][N] (S row slices, one thread per column), then the slices in order
__global__ __launch_bounds__(256) void colsum_f32_partial_kernel(const float* __restrict__ X, long ldx, int M, int N, int rows_per,
                                                                 float* __restrict__ part, const float* __restrict__ mask, long ldm) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = r0;
  if (mask != nullptr) {                    // x * (mask > 0): the bias gradient behind a ReLU
    for (; r < r1; ++r) s0 += mask[(long)r * ldm + n] > 0.f ? X[(long)r * ldx + n] : 0.f;
  } else {
    for (; r + 4 <= r1; r += 4) {
      s0 += X[(long)r * ldx + n]; s1 += X[(long)(r + 1) * ldx + n]; s2 += X[(long)(r + 2) * ldx + n]; s3 += X[(long)(r + 3) * ldx + n];
    }
    for (; r < r1; ++r) s0 += X[(long)r * ldx + n];
  }
  part[(long)blockIdx.y * N + n] = (s0 + s1) + (s2 + s3);
}
// out[seg][n] (+)= sum over p of part[p][seg * N + n]   (nseg segments side by side in every partial row)
struct SegOut { float* o[3]; };
__global__ __launch_bounds__(256) void colreduce_f32_kernel(const float* __restrict__ part, int np, int N, int nseg, SegOut out, int accumulate) {
  const int n = blockIdx.x * 256 + threadIdx.x, seg = blockIdx.y;
  if (n >= N) return;
  float s = 0.f;
  for (int p = 0; p < np; ++p) s += part[(long)p * nseg * N + (long)seg * N + n];
  float* o = out.o[seg] + n;
  *o = accumulate ? *o + s : s;
}
extern "C" long hriemo_colsum_f32_workspace_bytes(int M, int N) { (void)M; return 64L * N * 4; }
extern "C" int hriemo_colsum_f32(const float* X, long ldx, int M, int N, const float* mask, long ldmask, float* out, int accumulate,
                                 float* workspace, hipStream_t st) {
  HRIEMO_CHECK(M > 0 && N > 0 && out != nullptr && workspace != nullptr, "colsum_f32: bad arguments");
  int slices = (M + 63) / 64;
  if (slices > 64) slices = 64;
  const int rows_per = (M + slices - 1) / slices;
  slices = (M + rows_per - 1) / rows_per;
  hipLaunchKernelGGL(colsum_f32_partial_kernel, dim3((N + 255) / 256, slices), dim3(256), 0, st, X, ldx, M, N, rows_per, workspace, mask, ldmask);
  SegOut so; so.o[0] = out; so.o[1] = so.o[2] = nullptr;
  hipLaunchKernelGGL(colreduce_f32_kernel, dim3((N + 255) / 256, 1), dim3(256), 0, st, workspace, slices, N, 1, so, accumulate);
  HRIEMO_LAUNCH_CHECK("colsum_f32 kernels");
  return 0;
}

// ------------------------------------------------------------------------------------------- LayerNorm(x + g) backward, fp32
// dS = d loss / d (x + drop(g)) (= dX; = dG too without dropout, else dG = dS * keep / (1 - p) goes to its own matrix); per-block column partials of dgamma = sum dY * xhat, dbeta = sum dY and
// dbias = sum dG (the bias of the Linear that produced g).  The row statistics are recomputed (the row is in registers anyway).
template <int NV4>
__global__ __launch_bounds__(256) void add_ln_bwd_f32_kernel(const float* __restrict__ dY, const float* __restrict__ G, const float* __restrict__ X,
                                                             const float* __restrict__ gamma, float* __restrict__ dS, float* __restrict__ dG,
                                                             float* __restrict__ part, int M, int d, float eps, RowDrop dr) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;            // [3][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nq = d >> 2;
  const float invd = 1.f / (float)d;
  const uint32_t dkey = dr.thr16 != 0 ? site_key(eff_seed(dr.seed, dr.seed_dev), dr.site, 0u) : 0u;
  f32x4 ag[NV4], ab[NV4], abias[NV4];
#pragma unroll
  for (int c = 0; c < NV4; ++c) ag[c] = ab[c] = abias[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
    f32x4 s[NV4], dy[NV4];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NV4; ++c) {
      const int q = lane + 64 * c;
      s[c] = dy[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (q < nq) {
        s[c] = *(const f32x4*)(G + row * d + q * 4);
        if (dr.thr16 != 0) s[c] = drop4(s[c], dkey, (uint32_t)(row + dr.row_off), (uint32_t)(q * 4), dr.thr16, dr.inv_keep);
        if (X != nullptr) s[c] += *(const f32x4*)(X + row * d + q * 4);
        dy[c] = *(const f32x4*)(dY + row * d + q * 4);
        sum += (s[c][0] + s[c][1]) + (s[c][2] + s[c][3]);
      }
    }
    const float mu = wave_sum(sum) * invd;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NV4; ++c)
      if (lane + 64 * c < nq) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float t = s[c][j] - mu; sq += t * t; }
      }
    const float rstd = 1.f / sqrtf(wave_sum(sq) * invd + eps);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int c = 0; c < NV4; ++c) {
      const int q = lane + 64 * c;
      if (q < nq) {
        const f32x4 gm = *(const f32x4*)(gamma + q * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (s[c][j] - mu) * rstd;
          const float dyg = dy[c][j] * gm[j];
          c1 += dyg;
          c2 += dyg * xh;
          ag[c][j] += dy[c][j] * xh;
          ab[c][j] += dy[c][j];
          s[c][j] = xh;                 // xhat from here on
          dy[c][j] = dyg;               // dy * gamma from here on
        }
      }
    }
    c1 = wave_sum(c1) * invd;
    c2 = wave_sum(c2) * invd;
#pragma unroll
    for (int c = 0; c < NV4; ++c) {
      const int q = lane + 64 * c;
      if (q < nq) {
        f32x4 ds;
#pragma unroll
        for (int j = 0; j < 4; ++j) ds[j] = rstd * (dy[c][j] - c1 - s[c][j] * c2);
        *(f32x4*)(dS + row * d + q * 4) = ds;
        if (dr.thr16 != 0) {              // the dropped branch's gradient: dG = dS * (keep ? 1 / (1 - p) : 0); the bias sits inside g
          const f32x4 f = drop4_factor(dkey, (uint32_t)(row + dr.row_off), (uint32_t)(q * 4), dr.thr16, dr.inv_keep);
          ds *= f;
          *(f32x4*)(dG + row * d + q * 4) = ds;
        }
        abias[c] += ds;
      }
    }
  }
  // (one call per accumulator array: picking the array by a run-time index would put all three into scratch)
  auto fold = [&](const f32x4 (&a)[NV4], float* dst) {
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int c = 0; c < NV4; ++c) {
          const int q = lane + 64 * c;
          if (q < nq) {
            f32x4 v = a[c];
            if (w != 0) v += *(const f32x4*)(dst + q * 4);
            *(f32x4*)(dst + q * 4) = v;
          }
        }
      }
      __syncthreads();
    }
  };
  fold(ag, red);
  fold(ab, red + d);
  fold(abias, red + 2 * d);
  float* out = part + (long)blockIdx.x * 3 * d;
  for (int t = threadIdx.x; t < 3 * d; t += 256) out[t] = red[t];
}
static int add_ln_bwd_f32_blocks(int M) { int g = (M + 3) / 4; return g > 512 ? 512 : g; }
extern "C" long hriemo_add_ln_bwd_f32_workspace_bytes(int M, int d) { return (long)add_ln_bwd_f32_blocks(M) * 3 * d * 4; }
extern "C" int hriemo_add_ln_bwd_f32(const float* dY, const float* G, const float* X, const float* gamma, float* dS, float* dG, float* dgamma,
                                     float* dbeta, float* dbias, int accumulate, int M, int d, float eps, float p, uint64_t seed,
                                     const unsigned long long* seed_dev, uint32_t site, long row_off, float* workspace, hipStream_t st) {
  HRIEMO_CHECK(M > 0 && d > 0 && d % 4 == 0 && d <= 1024, "add_ln_bwd_f32: d=%d must be a multiple of 4, at most 1024", d);
  HRIEMO_CHECK(p >= 0.f && p < 1.f, "add_ln_bwd_f32: dropout p=%f", (double)p);
  const RowDrop dr = make_row_drop(p, seed, seed_dev, site, row_off);
  HRIEMO_CHECK(dr.thr16 == 0 || dG != nullptr, "add_ln_bwd_f32: with dropout the gradient of g differs from dS and needs its own output (dG)");
  HRIEMO_CHECK(dY != nullptr && G != nullptr && gamma != nullptr && dS != nullptr && dgamma != nullptr && dbeta != nullptr && workspace != nullptr,
               "add_ln_bwd_f32: missing operand");
  const int nb = add_ln_bwd_f32_blocks(M);
  hriemo_prof_begin(HP_ROWOPS, st);
  // (the row, its gradient and three column accumulators live in registers: 1024 columns -- every BASELINE config -- is what fits
  // without scratch; wider rows are refused above rather than served by a spilling instantiation)
  hipLaunchKernelGGL((add_ln_bwd_f32_kernel<4>), dim3(nb), dim3(256), 3 * d * 4, st, dY, G, X, gamma, dS, dG, workspace, M, d, eps, dr);
  HRIEMO_LAUNCH_CHECK("add_ln_bwd_f32_kernel");
  hriemo_prof_end(HP_ROWOPS, st, (double)M * d * 16.0);
  SegOut so; so.o[0] = dgamma; so.o[1] = dbeta; so.o[2] = dbias;
  hipLaunchKernelGGL(colreduce_f32_kernel, dim3((d + 255) / 256, dbias != nullptr ? 3 : 2), dim3(256), 0, st, workspace, nb, d, 3, so, accumulate);
  HRIEMO_LAUNCH_CHECK("colreduce_f32_kernel");
  return 0;
}

// ------------------------------------------------------------------------------------------- attention backward on the fp32 MFMA
// Two kernels, both in the forward kernel's register layouts (v_mfma_f32_16x16x4_f32, lane (i, g)), both deterministic:
//  dQ: block = 64 queries of one (batch, head), key tiles of 64 through LDS.  S^T = K.Q^T and dP^T = V.dO^T leave the
//      accumulators as [key 16n+4g+r][query i]; p = exp(s - lse), ds = p (dp - delta) with delta = rowsum(dO * O) (written out for
//      the second kernel); dQ^T += K^T . dS^T is the forward's O^T += V^T . P^T with K for V.
//  dK/dV: block = 64 keys, the keys on the lanes (K, V rows in registers), query tiles of 64 (Q, dO rows, lse, delta) through LDS:
//      S = Q.K^T and dP = dO.V^T leave [query 16n+4g+r][key i]; dV^T += dO^T . P and dK^T += Q^T . dS contract over the tile's queries.
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_f32_kernel(const float* __restrict__ Q, long ldq, const float* __restrict__ K, long ldk,
                                                              const float* __restrict__ V, long ldv, const float* __restrict__ O, long ldo,
                                                              const float* __restrict__ dO, long lddo, const uint8_t* __restrict__ kpm,
                                                              const float* __restrict__ lse, float* __restrict__ dQ, long lddq,
                                                              float* __restrict__ delta, int H, int Lq, int Lk, float scale, AttnDrop dr) {
  constexpr int LDR = HD + 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Ks = (float*)smem_raw;
  float* Vs = Ks + 64 * LDR;
  float* pad = Vs + 64 * LDR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int q = min(q0 + i, Lq - 1);
  const float* qp = Q + ((long)b * Lq + q) * ldq + h * HD;
  const float* dop = dO + ((long)b * Lq + q) * lddo + h * HD;
  const float* op = O + ((long)b * Lq + q) * ldo + h * HD;
  float qf[HD / 4], dof[HD / 4];
  float dl = 0.f;
#pragma unroll
  for (int ks = 0; ks < HD / 4; ++ks) {
    qf[ks] = qp[4 * ks + g] * scale;
    dof[ks] = dop[4 * ks + g];
    dl += dof[ks] * op[4 * ks + g];
  }
  dl += __shfl_xor(dl, 16);
  dl += __shfl_xor(dl, 32);
  const float ls = lse[((long)b * H + h) * Lq + q];
  if (g == 0 && q0 + i < Lq) delta[((long)b * H + h) * Lq + q0 + i] = dl;
  const uint32_t dkey = dr.thr16 != 0 ? site_key(eff_seed(dr.seed, dr.seed_dev), dr.site, (uint32_t)((dr.b_offset + b) * H + h)) : 0u;
  f32x4 dq[HD / 16];
#pragma unroll
  for (int t = 0; t < HD / 16; ++t) dq[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* Kb = K + (long)b * Lk * ldk + h * HD;
  const float* Vb = V + (long)b * Lk * ldv + h * HD;
  for (int k0 = 0; k0 < Lk; k0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * (HD / 4); e += 256) {
      const int r = e / (HD / 4), c = (e - r * (HD / 4)) * 4;
      const int key = k0 + r;
      f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (key < Lk) { kv = *(const f32x4*)(Kb + (long)key * ldk + c); vv = *(const f32x4*)(Vb + (long)key * ldv + c); }
      *(f32x4*)(Ks + r * LDR + c) = kv;
      *(f32x4*)(Vs + r * LDR + c) = vv;
    }
    if (tid < 64) {
      const int key = k0 + tid;
      pad[tid] = (key < Lk && (kpm == nullptr || kpm[(long)b * Lk + key] == 0)) ? 0.f : -INFINITY;
    }
    __syncthreads();
    f32x4 s[4], dp[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      s[n] = dp[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HD / 4; ++ks) {
        s[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(16 * n + i) * LDR + 4 * ks + g], qf[ks], s[n], 0, 0, 0);
        dp[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[(16 * n + i) * LDR + 4 * ks + g], dof[ks], dp[n], 0, 0, 0);
      }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = expf(s[n][r] + pad[16 * n + 4 * g + r] - ls);
        float dpv = dp[n][r];                     // gradient of the DROPPED weight; that of the weight itself is keep / (1 - p) times it
        if (dr.thr16 != 0) dpv = keep16(dkey, (uint32_t)(q0 + i), (uint32_t)(k0 + 16 * n + 4 * g + r), dr.thr16) ? dpv * dr.inv_keep : 0.f;
        s[n][r] = p * (dpv - dl);                 // dS^T[key][query]
      }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < HD / 16; ++t)
          dq[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(16 * n + 4 * g + r) * LDR + 16 * t + i], s[n][r], dq[t], 0, 0, 0);
  }
  if (q0 + i < Lq) {
    float* dqp = dQ + ((long)b * Lq + q0 + i) * lddq + h * HD;
#pragma unroll
    for (int t = 0; t < HD / 16; ++t) {
      f32x4 v = dq[t];
      v *= scale;
      *(f32x4*)(dqp + 16 * t + 4 * g) = v;
    }
  }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_f32_kernel(const float* __restrict__ Q, long ldq, const float* __restrict__ K, long ldk,
                                                               const float* __restrict__ V, long ldv, const float* __restrict__ dO, long lddo,
                                                               const uint8_t* __restrict__ kpm, const float* __restrict__ lse,
                                                               const float* __restrict__ delta, float* __restrict__ dK, long lddk,
                                                               float* __restrict__ dV, long lddv, int H, int Lq, int Lk, float scale,
                                                               AttnDrop dr) {
  constexpr int LDR = HD + 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Qs = (float*)smem_raw;
  float* Ds = Qs + 64 * LDR;
  float* lss = Ds + 64 * LDR;          // [64] lse of the tile's queries (+inf past the end: p = 0)
  float* dls = lss + 64;               // [64] delta
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int k0 = blockIdx.x * 64 + wave * 16;
  const int key = min(k0 + i, Lk - 1);
  const bool kvalid = (k0 + i < Lk) && (kpm == nullptr || kpm[(long)b * Lk + key] == 0);
  const float kpad = kvalid ? 0.f : -INFINITY;
  const uint32_t dkey = dr.thr16 != 0 ? site_key(eff_seed(dr.seed, dr.seed_dev), dr.site, (uint32_t)((dr.b_offset + b) * H + h)) : 0u;
  const float* kp = K + ((long)b * Lk + key) * ldk + h * HD;
  const float* vp = V + ((long)b * Lk + key) * ldv + h * HD;
  float kf[HD / 4], vf[HD / 4];
#pragma unroll
  for (int ks = 0; ks < HD / 4; ++ks) { kf[ks] = kp[4 * ks + g] * scale; vf[ks] = vp[4 * ks + g]; }
  f32x4 dk[HD / 16], dv[HD / 16];
#pragma unroll
  for (int t = 0; t < HD / 16; ++t) dk[t] = dv[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* Qb = Q + (long)b * Lq * ldq + h * HD;
  const float* Db = dO + (long)b * Lq * lddo + h * HD;
  for (int q0 = 0; q0 < Lq; q0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * (HD / 4); e += 256) {
      const int r = e / (HD / 4), c = (e - r * (HD / 4)) * 4;
      const int qq = q0 + r;
      f32x4 qv = {0.f, 0.f, 0.f, 0.f}, dv4 = {0.f, 0.f, 0.f, 0.f};
      if (qq < Lq) { qv = *(const f32x4*)(Qb + (long)qq * ldq + c); dv4 = *(const f32x4*)(Db + (long)qq * lddo + c); }
      *(f32x4*)(Qs + r * LDR + c) = qv;
      *(f32x4*)(Ds + r * LDR + c) = dv4;
    }
    if (tid < 64) {
      const int qq = q0 + tid;
      lss[tid] = qq < Lq ? lse[((long)b * H + h) * Lq + qq] : INFINITY;
      dls[tid] = qq < Lq ? delta[((long)b * H + h) * Lq + qq] : 0.f;
    }
    __syncthreads();
    f32x4 s[4], dp[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      s[n] = dp[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HD / 4; ++ks) {
        s[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[(16 * n + i) * LDR + 4 * ks + g], kf[ks], s[n], 0, 0, 0);
        dp[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ds[(16 * n + i) * LDR + 4 * ks + g], vf[ks], dp[n], 0, 0, 0);
      }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qr = 16 * n + 4 * g + r;
        const float p = expf(s[n][r] + kpad - lss[qr]);       // [query qr][key i]
        const float f = dr.thr16 == 0 ? 1.f : (keep16(dkey, (uint32_t)(q0 + qr), (uint32_t)(k0 + i), dr.thr16) ? dr.inv_keep : 0.f);
        dp[n][r] = p * (f * dp[n][r] - dls[qr]);              // dS
        s[n][r] = f * p;                                      // the dropped weight: what multiplied V in the forward
      }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < HD / 16; ++t) {
          dv[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ds[(16 * n + 4 * g + r) * LDR + 16 * t + i], s[n][r], dv[t], 0, 0, 0);
          dk[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[(16 * n + 4 * g + r) * LDR + 16 * t + i], dp[n][r], dk[t], 0, 0, 0);
        }
  }
  if (k0 + i < Lk) {
    float* dkp = dK + ((long)b * Lk + k0 + i) * lddk + h * HD;
    float* dvp = dV + ((long)b * Lk + k0 + i) * lddv + h * HD;
#pragma unroll
    for (int t = 0; t < HD / 16; ++t) {
      f32x4 v = dk[t];
      v *= scale;                                   // S = scale Q.K^T: dK = scale . dS^T . Q (Q sits unscaled in LDS)
      *(f32x4*)(dkp + 16 * t + 4 * g) = v;
      *(f32x4*)(dvp + 16 * t + 4 * g) = dv[t];
    }
  }
}


// ------------------------------------------------------------------------------------------- gate backward pieces (fp32)
// h = w * A[:, :L] + (1 - w) * T[:, :L], beta = mean_d(w), w = sigmoid(pre)  (beta_gate_tacfn.py:92-116):
//   dpre[b][c] = (sum_{l < L} dH[b][l][c] * (A - T)[b][l][c] + dbeta[b] / d) * w (1 - w)
__global__ __launch_bounds__(256) void gate_dpre_f32_kernel(const float* __restrict__ dH, const float* __restrict__ A, int La,
                                                            const float* __restrict__ T, int Lt, const float* __restrict__ w,
                                                            const float* __restrict__ dbeta, float* __restrict__ dpre, int L, int d) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  float s0 = 0.f, s1 = 0.f;
  int l = 0;
  for (; l + 2 <= L; l += 2) {
    s0 += dH[((long)b * L + l) * d + c] * (A[((long)b * La + l) * d + c] - T[((long)b * Lt + l) * d + c]);
    s1 += dH[((long)b * L + l + 1) * d + c] * (A[((long)b * La + l + 1) * d + c] - T[((long)b * Lt + l + 1) * d + c]);
  }
  for (; l < L; ++l) s0 += dH[((long)b * L + l) * d + c] * (A[((long)b * La + l) * d + c] - T[((long)b * Lt + l) * d + c]);
  const float wv = w[(long)b * d + c];
  const float dw = (s0 + s1) + (dbeta != nullptr ? dbeta[b] / (float)d : 0.f);
  dpre[(long)b * d + c] = dw * wv * (1.f - wv);
}
// gate input [a, t, |a - t|, a * t] (:87-89): da = g0 + sign(a - t) g2 + t g3, dt = g1 - sign(a - t) g2 + a g3   (sign(0) = 0, as torch.abs)
__global__ __launch_bounds__(256) void gate_in_bwd_f32_kernel(const float* __restrict__ dgin, const float* __restrict__ a, const float* __restrict__ t,
                                                              float* __restrict__ da, float* __restrict__ dt, int d) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const float av = a[(long)b * d + c], tv = t[(long)b * d + c];
  const float* g = dgin + (long)b * 4 * d;
  const float sg = av > tv ? 1.f : (av < tv ? -1.f : 0.f);
  da[(long)b * d + c] = g[c] + sg * g[2 * d + c] + tv * g[3 * d + c];
  dt[(long)b * d + c] = g[d + c] - sg * g[2 * d + c] + av * g[3 * d + c];
}
// gradient that reaches the gate's LayerNorm output of one modality, row (b, l) of Lx:
//   dY = (l < L ? wsel * dH[b][l] : 0) + (valid(b, l) ? dpool[b] / max(#valid(b), 1) : 0),  wsel = w (audio) or 1 - w (text)
__global__ __launch_bounds__(256) void gate_dy_f32_kernel(const float* __restrict__ dH, const float* __restrict__ w, int is_a,
                                                          const float* __restrict__ dpool, const uint8_t* __restrict__ mask,
                                                          float* __restrict__ dY, int L, int Lx, int d) {
  __shared__ float cnt_s;
  const int b = blockIdx.y, l = blockIdx.x, tid = threadIdx.x;
  if (tid < 64) {
    float n = 0.f;
    for (int k = tid; k < Lx; k += 64) n += (mask == nullptr || mask[(long)b * Lx + k] == 0) ? 1.f : 0.f;
    n = wave_sum(n);
    if (tid == 0) cnt_s = fmaxf(n, 1.f);
  }
  __syncthreads();
  const bool valid = mask == nullptr || mask[(long)b * Lx + l] == 0;
  const float inv = valid ? 1.f / cnt_s : 0.f;
  for (int c = tid; c < d; c += 256) {
    const float wv = w[(long)b * d + c];
    float v = dpool[(long)b * d + c] * inv;
    if (l < L) v += (is_a ? wv : 1.f - wv) * dH[((long)b * L + l) * d + c];
    dY[((long)b * Lx + l) * d + c] = v;
  }
}
// logits[m] = z[m] . w + b (emotion_decoder.py:155): dz = dl w, dw = sum_m dl[m] z[m], db = sum_m dl[m]; fixed summation order
__global__ __launch_bounds__(256) void rowdot_bwd_f32_kernel(const float* __restrict__ dl, const float* __restrict__ Z, const float* __restrict__ w,
                                                             float* __restrict__ dZ, float* __restrict__ dw, float* __restrict__ db, int M, int d,
                                                             int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < d) {
    const float wc = w[c];
    float acc = 0.f;
    for (int r = 0; r < M; ++r) {
      const float gr = dl[r];
      acc += gr * Z[(long)r * d + c];
      dZ[(long)r * d + c] = gr * wc;
    }
    dw[c] = accumulate ? dw[c] + acc : acc;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float t = 0.f;
    for (int r = 0; r < M; ++r) t += dl[r];
    db[0] = accumulate ? db[0] + t : t;
  }
}

extern "C" int hriemo_gate_dpre_f32(const float* dH, const float* A, int La, const float* T, int Lt, const float* w, const float* dbeta,
                                    float* dpre, int B, int L, int d, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && L > 0 && d > 0 && L <= La && L <= Lt, "gate_dpre_f32: bad shape");
  hipLaunchKernelGGL(gate_dpre_f32_kernel, dim3((d + 255) / 256, B), dim3(256), 0, st, dH, A, La, T, Lt, w, dbeta, dpre, L, d);
  HRIEMO_LAUNCH_CHECK("gate_dpre_f32_kernel");
  return 0;
}
extern "C" int hriemo_gate_input_bwd_f32(const float* dgin, const float* a_pool, const float* t_pool, float* da, float* dt, int B, int d,
                                         hipStream_t st) {
  HRIEMO_CHECK(B > 0 && d > 0, "gate_input_bwd_f32: empty input");
  hipLaunchKernelGGL(gate_in_bwd_f32_kernel, dim3((d + 255) / 256, B), dim3(256), 0, st, dgin, a_pool, t_pool, da, dt, d);
  HRIEMO_LAUNCH_CHECK("gate_in_bwd_f32_kernel");
  return 0;
}
extern "C" int hriemo_gate_dy_f32(const float* dH, const float* w, int is_a, const float* dpool, const unsigned char* mask, float* dY, int B,
                                  int L, int Lx, int d, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && L > 0 && Lx >= L && d > 0, "gate_dy_f32: bad shape");
  hipLaunchKernelGGL(gate_dy_f32_kernel, dim3(Lx, B), dim3(256), 0, st, dH, w, is_a, dpool, mask, dY, L, Lx, d);
  HRIEMO_LAUNCH_CHECK("gate_dy_f32_kernel");
  return 0;
}
extern "C" int hriemo_rowdot_bwd_f32(const float* dl, const float* Z, const float* w, float* dZ, float* dw, float* db, int accumulate, int M,
                                     int d, hipStream_t st) {
  HRIEMO_CHECK(M > 0 && d > 0, "rowdot_bwd_f32: empty input");
  hipLaunchKernelGGL(rowdot_bwd_f32_kernel, dim3((d + 255) / 256), dim3(256), 0, st, dl, Z, w, dZ, dw, db, M, d, accumulate);
  HRIEMO_LAUNCH_CHECK("rowdot_bwd_f32_kernel");
  return 0;
}

#define DISPATCH_HD_F32(hd, CALL)     \
  switch (hd) {                       \
    case 16: { CALL(16); } break;     \
    case 32: { CALL(32); } break;     \
    case 64: { CALL(64); } break;     \
    case 96: { CALL(96); } break;     \
    case 128: { CALL(128); } break;   \
    default: hriemo_set_error("attn_f32: head_dim=%d is not built (16, 32, 64, 96, 128)", hd); return 1; \
  }

extern "C" int hriemo_attn_fwd_f32(const float* Q, long ldq, const float* K, long ldk, const float* V, long ldv, float* O, long ldo,
                                   const unsigned char* key_padding_mask, float* lse, int B, int H, int Lq, int Lk, int head_dim,
                                   float p, uint64_t seed, const unsigned long long* seed_dev, uint32_t site, int b_offset, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && H > 0 && Lq > 0 && Lk > 0, "attn_fwd_f32: empty problem");
  HRIEMO_CHECK(p >= 0.f && p < 1.f, "attn_fwd_f32: dropout p=%f", (double)p);
  const AttnDrop dr = make_attn_drop(p, seed, seed_dev, site, b_offset);
  HRIEMO_CHECK(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)Q % 16) == 0 && ((uintptr_t)K % 16) == 0 &&
                   ((uintptr_t)V % 16) == 0 && ((uintptr_t)O % 16) == 0, "attn_fwd_f32: operands must be 16-byte aligned");
  const float scale = 1.0f / sqrtf((float)head_dim);
  const dim3 grid((Lq + 63) / 64, B * H);
  hriemo_prof_begin(HP_ATTN_FWD, st);
#define CALL(HD)                                                                                                                         \
  {                                                                                                                                      \
    const int lds = (2 * 64 * (HD + 4) + 64) * 4;                                                                                        \
    static bool attr = false;                                                                                                            \
    if (!attr) { hipFuncSetAttribute((const void*)attn_fwd_f32_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; } \
    hipLaunchKernelGGL((attn_fwd_f32_kernel<HD>), grid, dim3(256), lds, st, Q, ldq, K, ldk, V, ldv, O, ldo, key_padding_mask, lse, H, Lq, Lk, scale, dr); \
  }
  DISPATCH_HD_F32(head_dim, CALL)
#undef CALL
  HRIEMO_LAUNCH_CHECK("attn_fwd_f32_kernel");
  hriemo_prof_end(HP_ATTN_FWD, st, 4.0 * B * H * (double)Lq * Lk * head_dim);
  return 0;
}

extern "C" int hriemo_attn_probs_f32(const float* Q, long ldq, const float* K, long ldk, const unsigned char* key_padding_mask,
                                     const float* lse, float* probs, int B, int H, int Lq, int Lk, int head_dim, float p, uint64_t seed,
                                     const unsigned long long* seed_dev, uint32_t site, int b_offset, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && H > 0 && Lq > 0 && Lk > 0 && lse != nullptr && probs != nullptr, "attn_probs_f32: empty problem");
  HRIEMO_CHECK(p >= 0.f && p < 1.f, "attn_probs_f32: dropout p=%f", (double)p);
  const AttnDrop dr = make_attn_drop(p, seed, seed_dev, site, b_offset);
  HRIEMO_CHECK(ldq % 4 == 0 && ldk % 4 == 0 && ((uintptr_t)Q % 16) == 0 && ((uintptr_t)K % 16) == 0, "attn_probs_f32: operands must be 16-byte aligned");
  const float scale = 1.0f / sqrtf((float)head_dim);
  const dim3 grid((Lq + 63) / 64, B);
#define CALL(HD)                                                                                                                           \
  {                                                                                                                                        \
    const int lds = (64 * (HD + 4) + 64) * 4;                                                                                              \
    hipLaunchKernelGGL((attn_probs_f32_kernel<HD>), grid, dim3(256), lds, st, Q, ldq, K, ldk, key_padding_mask, lse, probs, H, Lq, Lk, scale, dr); \
  }
  DISPATCH_HD_F32(head_dim, CALL)
#undef CALL
  HRIEMO_LAUNCH_CHECK("attn_probs_f32_kernel");
  return 0;
}

extern "C" int hriemo_attn_bwd_f32(const float* Q, long ldq, const float* K, long ldk, const float* V, long ldv, const float* O, long ldo,
                                   const float* dO, long lddo, const unsigned char* key_padding_mask, const float* lse, float* dQ, long lddq,
                                   float* dK, long lddk, float* dV, long lddv, float* delta, int B, int H, int Lq, int Lk, int head_dim,
                                   float p, uint64_t seed, const unsigned long long* seed_dev, uint32_t site, int b_offset, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && H > 0 && Lq > 0 && Lk > 0 && lse != nullptr && delta != nullptr, "attn_bwd_f32: empty problem");
  HRIEMO_CHECK(p >= 0.f && p < 1.f, "attn_bwd_f32: dropout p=%f", (double)p);
  const AttnDrop dr = make_attn_drop(p, seed, seed_dev, site, b_offset);
  HRIEMO_CHECK(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && lddo % 4 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0,
               "attn_bwd_f32: leading dimensions must be multiples of 4");
  HRIEMO_CHECK(((uintptr_t)Q % 16) == 0 && ((uintptr_t)K % 16) == 0 && ((uintptr_t)V % 16) == 0 && ((uintptr_t)O % 16) == 0 && ((uintptr_t)dO % 16) == 0 &&
                   ((uintptr_t)dQ % 16) == 0 && ((uintptr_t)dK % 16) == 0 && ((uintptr_t)dV % 16) == 0, "attn_bwd_f32: operands must be 16-byte aligned");
  const float scale = 1.0f / sqrtf((float)head_dim);
  hriemo_prof_begin(HP_ATTN_BWD_DQ, st);
#define CALL(HD)                                                                                                                          \
  {                                                                                                                                       \
    const int lds = (2 * 64 * (HD + 4) + 64) * 4;                                                                                         \
    static bool attr = false;                                                                                                             \
    if (!attr) {                                                                                                                          \
      hipFuncSetAttribute((const void*)attn_bwd_dq_f32_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                      \
      hipFuncSetAttribute((const void*)attn_bwd_dkv_f32_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds + 256);               \
      attr = true;                                                                                                                        \
    }                                                                                                                                     \
    hipLaunchKernelGGL((attn_bwd_dq_f32_kernel<HD>), dim3((Lq + 63) / 64, B * H), dim3(256), lds, st, Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, \
                       key_padding_mask, lse, dQ, lddq, delta, H, Lq, Lk, scale, dr);                                                    \
    hipLaunchKernelGGL((attn_bwd_dkv_f32_kernel<HD>), dim3((Lk + 63) / 64, B * H), dim3(256), lds + 256, st, Q, ldq, K, ldk, V, ldv, dO, lddo, \
                       key_padding_mask, lse, delta, dK, lddk, dV, lddv, H, Lq, Lk, scale, dr);                                          \
  }
  DISPATCH_HD_F32(head_dim, CALL)
#undef CALL
  HRIEMO_LAUNCH_CHECK("attn_bwd_f32 kernels");
  hriemo_prof_end(HP_ATTN_BWD_DQ, st, 14.0 * B * H * (double)Lq * Lk * head_dim);
  return 0;
}
