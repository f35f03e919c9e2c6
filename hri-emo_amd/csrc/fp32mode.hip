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
// Forward only (no dropout): training stays on the bf16 path.
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

// ------------------------------------------------------------------------------------------- LayerNorm(x + g), fp32 in and out
// one wave per row, the row in registers (d <= 64 * 4 * NV4)
template <int NV4>
__global__ __launch_bounds__(256) void add_ln_f32_kernel(const float* __restrict__ G, const float* __restrict__ X, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ Y32, bf16_t* __restrict__ Y16, int M,
                                                         int d, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nq = d >> 2;
  f32x4 s[NV4];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NV4; ++c) {
    const int q = lane + 64 * c;
    s[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (q < nq) {
      s[c] = *(const f32x4*)(G + row * d + q * 4);
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
                                 float eps, hipStream_t st) {
  HRIEMO_CHECK(M > 0 && d > 0 && d % 4 == 0 && d <= 4096, "add_ln_f32: d=%d must be a multiple of 4, at most 4096", d);
  hriemo_prof_begin(HP_ROWOPS, st);
  const dim3 grid((M + 3) / 4);
  if (d <= 1024) hipLaunchKernelGGL((add_ln_f32_kernel<4>), grid, dim3(256), 0, st, G, X, gamma, beta, Y32, (bf16_t*)Y16, M, d, eps);
  else hipLaunchKernelGGL((add_ln_f32_kernel<16>), grid, dim3(256), 0, st, G, X, gamma, beta, Y32, (bf16_t*)Y16, M, d, eps);
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
                                                           float scale) {
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
        s[n][r] = p;
        lrun += p;
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
  const float inv = 1.f / l;
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
                                                             float* __restrict__ probs, int H, int Lq, int Lk, float scale) {
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
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < HD / 4; ++ks)
          s = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(16 * n + i) * LDR + 4 * ks + g], qp[4 * ks + g] * scale, s, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[n][r] += expf(s[r] + pad[16 * n + 4 * g + r] - ls);     // lse = -inf (no valid key): NaN, as PyTorch
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
                                   hipStream_t st) {
  HRIEMO_CHECK(B > 0 && H > 0 && Lq > 0 && Lk > 0, "attn_fwd_f32: empty problem");
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
    hipLaunchKernelGGL((attn_fwd_f32_kernel<HD>), grid, dim3(256), lds, st, Q, ldq, K, ldk, V, ldv, O, ldo, key_padding_mask, lse, H, Lq, Lk, scale); \
  }
  DISPATCH_HD_F32(head_dim, CALL)
#undef CALL
  HRIEMO_LAUNCH_CHECK("attn_fwd_f32_kernel");
  hriemo_prof_end(HP_ATTN_FWD, st, 4.0 * B * H * (double)Lq * Lk * head_dim);
  return 0;
}

extern "C" int hriemo_attn_probs_f32(const float* Q, long ldq, const float* K, long ldk, const unsigned char* key_padding_mask,
                                     const float* lse, float* probs, int B, int H, int Lq, int Lk, int head_dim, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && H > 0 && Lq > 0 && Lk > 0 && lse != nullptr && probs != nullptr, "attn_probs_f32: empty problem");
  HRIEMO_CHECK(ldq % 4 == 0 && ldk % 4 == 0 && ((uintptr_t)Q % 16) == 0 && ((uintptr_t)K % 16) == 0, "attn_probs_f32: operands must be 16-byte aligned");
  const float scale = 1.0f / sqrtf((float)head_dim);
  const dim3 grid((Lq + 63) / 64, B);
#define CALL(HD)                                                                                                                           \
  {                                                                                                                                        \
    const int lds = (64 * (HD + 4) + 64) * 4;                                                                                              \
    hipLaunchKernelGGL((attn_probs_f32_kernel<HD>), grid, dim3(256), lds, st, Q, ldq, K, ldk, key_padding_mask, lse, probs, H, Lq, Lk, scale); \
  }
  DISPATCH_HD_F32(head_dim, CALL)
#undef CALL
  HRIEMO_LAUNCH_CHECK("attn_probs_f32_kernel");
  return 0;
}
