// HBM-bound row kernels of the fusion path (one wave64 per [d]-row, 16-byte bf16 vectors, fp32 math):
//   add_ln_fwd/bwd   y = LayerNorm(x + dropout(g))          cross_modal_block_tacfn.py:81,92,105,106,118,119
//                                                          emotion_decoder.py:43,55,59
//   ln_pool_fwd/bwd  LayerNorm + masked mean-pool           beta_gate_tacfn.py:6-24,79-84
//   gate_* / fuse_*  gate input, sigmoid gate, gated fuse   beta_gate_tacfn.py:87-116
//   colsum, cast, dropout, expand, rowdot                  bias grads, bf16 shadows, decoder glue
// Column reductions (dgamma/dbeta/dbias, pooling) are two-stage (per-block partials in a caller
// workspace, then a fixed-order reduce) so results are bitwise reproducible.
#include "common.h"
#include <string.h>
#include <stdlib.h>

#define EPS_DEFAULT 1e-5f

// rowmap (optional): the row index that keys the dropout hash, for rows that were gathered from a larger layout (packed varlen
// sequences: packed row -> row of the padded [B*L] layout), so that a packed launch drops exactly the elements the padded one drops
struct RowDrop { uint64_t seed; const unsigned long long* seed_dev; uint32_t site, thr16; float inv_keep; const long long* rowmap; };
static RowDrop row_drop(float p, uint64_t seed, const unsigned long long* seed_dev, uint32_t site) {
  DropCfg d = make_drop(p, seed, site);
  RowDrop r; r.seed = seed; r.seed_dev = seed_dev; r.site = site; r.thr16 = d.thr16; r.inv_keep = d.inv_keep; r.rowmap = nullptr;
  return r;
}
__device__ __forceinline__ uint32_t row_key(const RowDrop& dr) {
  return dr.thr16 != 0 ? site_key(eff_seed(dr.seed, dr.seed_dev), dr.site, 0u) : 0u;
}

// deterministic accumulate of per-wave column partials into an LDS row, wave by wave
template <int NCH>
__device__ __forceinline__ void block_colsum(float* lds_row, const float (&acc)[NCH][8], int nchunk, int lane, int wave, int nwave) {
  for (int w = 0; w < nwave; ++w) {
    if (wave == w) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (w == 0) lds_row[ch * 8 + j] = acc[c][j];
            else lds_row[ch * 8 + j] += acc[c][j];
          }
        }
      }
    }
    __syncthreads();
  }
}

// keep flags of the 8 columns of chunk `ch` in row `row`: four hashes, two columns each
__device__ __forceinline__ void row_keep8(uint32_t key32, uint32_t thr16, uint32_t row, int ch, bool (&kp)[8]) {
  if (thr16 == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) kp[j] = true;
    return;
  }
  const uint32_t hb = drop_base(key32, row, (uint32_t)(ch * 4));
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) {
    const uint32_t x = mix24(hb + (uint32_t)pr * DROP_CB);
    kp[2 * pr] = keep_lo(x, thr16);
    kp[2 * pr + 1] = keep_hi(x, thr16);
  }
}

// residual operand: the fp32 twin of the stream when given, else its bf16 copy, else 0
__device__ __forceinline__ void load_resid(const bf16_t* X, const float* X32, long off, float* xf) {
  if (X32 != nullptr) {
    const f32x4 a = *(const f32x4*)(X32 + off), b = *(const f32x4*)(X32 + off + 4);
    xf[0] = a[0]; xf[1] = a[1]; xf[2] = a[2]; xf[3] = a[3]; xf[4] = b[0]; xf[5] = b[1]; xf[6] = b[2]; xf[7] = b[3];
  } else if (X != nullptr) {
    bf8_to_f32(*(const bf16x8*)(X + off), xf);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) xf[j] = 0.f;
  }
}

// ------------------------------------------------------------------ y = LN(x + drop(g))
// per-column constants (LN gain / bias) kept in registers for the whole row loop (NCH <= 4; wider rows reload them)
template <int NCH>
__device__ __forceinline__ void load_cols(const float* __restrict__ v, int nchunk, int lane, float (&out)[NCH][8]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + 64 * c;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[c][j] = 0.f;
    if (ch < nchunk) {
      const f32x4 a = *(const f32x4*)(v + ch * 8), b = *(const f32x4*)(v + ch * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { out[c][e] = a[e]; out[c][4 + e] = b[e]; }
    }
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const bf16_t* __restrict__ G, const bf16_t* __restrict__ X,
                                                         const float* __restrict__ X32, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, bf16_t* __restrict__ Y,
                                                         float* __restrict__ Y32, float* __restrict__ mean_o,
                                                         float* __restrict__ rstd_o, int M, int d, float eps, RowDrop dr,
                                                         long row_offset, uint8_t* __restrict__ Yq, uint8_t* __restrict__ SY, long ldsy) {
  const uint32_t key32 = row_key(dr);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  const float invd = 1.f / (float)d;
  constexpr bool HOIST = NCH <= 4;
  float gm8[HOIST ? NCH : 1][8], bt8[HOIST ? NCH : 1][8];
  if (HOIST) { load_cols<HOIST ? NCH : 1>(gamma, d >> 3, threadIdx.x & 63, gm8); load_cols<HOIST ? NCH : 1>(beta, d >> 3, threadIdx.x & 63, bt8); }
  for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
    float s[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        float gf[8], xf[8];
        bf8_to_f32(*(const bf16x8*)(G + row * d + ch * 8), gf);
        load_resid(X, X32, row * d + ch * 8, xf);
        bool kp8[8];
        row_keep8(key32, dr.thr16, (uint32_t)(row_offset + (dr.rowmap != nullptr ? (long)dr.rowmap[row] : (long)row)), ch, kp8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gv = kp8[j] ? gf[j] * dr.inv_keep : 0.f;
          s[c][j] = xf[j] + gv;
          sum += s[c][j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) s[c][j] = 0.f;
      }
    }
    const float mu = wave_sum(sum) * invd;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (lane + 64 * c < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = s[c][j] - mu; sq += t * t; }
      }
    const float rstd = rsqrtf(wave_sum(sq) * invd + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          o[j] = (s[c][j] - mu) * rstd * (HOIST ? gm8[HOIST ? c : 0][j] : gamma[ch * 8 + j]) + (HOIST ? bt8[HOIST ? c : 0][j] : beta[ch * 8 + j]);
        const bf16x8 ob = f32_to_bf8(o);
        *(bf16x8*)(Y + row * d + ch * 8) = ob;
        if (Y32 != nullptr) {
          *(f32x4*)(Y32 + row * d + ch * 8) = (f32x4){o[0], o[1], o[2], o[3]};
          *(f32x4*)(Y32 + row * d + ch * 8 + 4) = (f32x4){o[4], o[5], o[6], o[7]};
        }
        if (Yq != nullptr) bf8_to_f32(ob, s[c]);      // the MX-fp8 copy quantises the bf16 output (what a separate pass would read)
      }
    }
    if (Yq != nullptr) {                              // kernel-uniform: MX-fp8 copy of y for the next GEMM (d % 32 == 0)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch >= nchunk) {
#pragma unroll
          for (int j = 0; j < 8; ++j) s[c][j] = 0.f;
        }
        typedef __attribute__((ext_vector_type(2))) int i32x2;
        int e;
        const i32x2 w = mx8_block(s[c], e);
        if (ch < nchunk) {
          *(i32x2*)(Yq + row * d + ch * 8) = w;
          if ((lane & 3) == 0) SY[(long)(ch >> 2) * ldsy + row] = (uint8_t)e;
        }
      }
    }
    if (lane == 0) { mean_o[row] = mu; rstd_o[row] = rstd; }
  }
}

// backward: dS (residual grad), dG = dS * mask/(1-p); column partials of dgamma, dbeta, dbias(=colsum dG)
template <int NCH>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ G,
                                                         const bf16_t* __restrict__ X, const float* __restrict__ X32,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                         bf16_t* __restrict__ dX, bf16_t* __restrict__ dG,
                                                         float* __restrict__ partials, int M, int d, RowDrop dr,
                                                         long row_offset) {
  const uint32_t key32 = row_key(dr);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;   // [3][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  const float invd = 1.f / (float)d;
  float ag[NCH][8], ab[NCH][8], abias[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) { ag[c][j] = 0.f; ab[c][j] = 0.f; abias[c][j] = 0.f; }

  constexpr bool HOIST = NCH <= 4;
  float gm8[HOIST ? NCH : 1][8];
  if (HOIST) load_cols<HOIST ? NCH : 1>(gamma, d >> 3, threadIdx.x & 63, gm8);
  for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
    const float mu = mean_i[row], rstd = rstd_i[row];
    float xh[NCH][8], dyg[NCH][8];
    bool kp[NCH][8];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        float gf[8], xf[8], dyf[8];
        bf8_to_f32(*(const bf16x8*)(G + row * d + ch * 8), gf);
        load_resid(X, X32, row * d + ch * 8, xf);
        bf8_to_f32(*(const bf16x8*)(dY + row * d + ch * 8), dyf);
        row_keep8(key32, dr.thr16, (uint32_t)(row_offset + (dr.rowmap != nullptr ? (long)dr.rowmap[row] : (long)row)), ch, kp[c]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gv = kp[c][j] ? gf[j] * dr.inv_keep : 0.f;
          const float sv = xf[j] + gv;
          xh[c][j] = (sv - mu) * rstd;
          dyg[c][j] = dyf[j] * (HOIST ? gm8[HOIST ? c : 0][j] : gamma[ch * 8 + j]);
          c1 += dyg[c][j];
          c2 += dyg[c][j] * xh[c][j];
          ag[c][j] += dyf[j] * xh[c][j];
          ab[c][j] += dyf[j];
        }
      }
    }
    c1 = wave_sum(c1) * invd;
    c2 = wave_sum(c2) * invd;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        float ds[8], dg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ds[j] = rstd * (dyg[c][j] - c1 - xh[c][j] * c2);
          dg[j] = kp[c][j] ? ds[j] * dr.inv_keep : 0.f;
          abias[c][j] += dg[j];
        }
        if (dX != nullptr) *(bf16x8*)(dX + row * d + ch * 8) = f32_to_bf8(ds);
        if (dG != nullptr) *(bf16x8*)(dG + row * d + ch * 8) = f32_to_bf8(dg);
      }
    }
  }
  block_colsum<NCH>(red, ag, nchunk, lane, wave, 4);
  block_colsum<NCH>(red + d, ab, nchunk, lane, wave, 4);
  block_colsum<NCH>(red + 2 * d, abias, nchunk, lane, wave, 4);
  float* out = partials + (long)blockIdx.x * 3 * d;
  for (int t = threadIdx.x; t < 3 * d; t += 256) out[t] = red[t];
}

// ---- round 4: the same two kernels on a QUAD mapping (d = 256 * NQ <= 1024, fp32 twin in and out, no MX copy) -------------
// The chunk mapping above gives lane l the 8-element chunks l, l + 64, ...: at d = 768 (96 chunks) lanes 32..63 idle on the second
// chunk and every lane carries registers for two.  Here lane l owns the 4-element quads l, l + 64, ...: d = 768 is exactly three
// quads on every lane (d = 1024: four), fp32 operands move as 16 bytes per lane and bf16 ones as 8, every wave-instruction a
// contiguous run of the row.  The loop is software-pipelined -- a wave issues the NEXT row's loads before it starts the current
// row's reductions, so two rows of a wave are in flight instead of one (what took ln_pool_fwd from 37 to 30 us in round 3) -- and
// written WITHOUT branches around memory operations: hipcc's wait insertion merges the pending-load state of the paths that
// meet at a join conservatively, and one conditional load or store in the loop turned the counted wait in front of the first
// use of the prefetched row into s_waitcnt vmcnt(0) (found in the ISA), which waits for the prefetch just issued.  So
//   * the mapping is built for rows that are whole multiples of 256 elements only (d = 256, 512, 768, 1024: all 64 lanes hold
//     NQ full quads, every address is one per-lane base + immediates); other widths keep the chunk-mapped kernels;
//   * the row after the last one is clamped onto row M - 1 (loaded, never consumed);
//   * per-row scalars (mean, rstd, the packed-row -> padded-row map) are read with the row index in an SGPR, i.e. as scalar
//     loads, which count on lgkmcnt and leave the vector-memory counter to the row data alone.
// The forward runs on a persistent grid (what the chip holds at once) instead of 1-2 rows per wave and a block turnover per
// row.  Same dropout stream as the chunk mapping (the hash is keyed by (row, column pair)); results differ from it only by the
// order of the two wave sums.
__device__ __forceinline__ void row_keep4(uint32_t key32, uint32_t thr16, uint32_t row, int q, bool (&kp)[4]) {
  const uint32_t hb = drop_base(key32, row, (uint32_t)(q * 2));
  const uint32_t x0 = mix24(hb), x1 = mix24(hb + DROP_CB);
  kp[0] = thr16 == 0 || keep_lo(x0, thr16); kp[1] = thr16 == 0 || keep_hi(x0, thr16);
  kp[2] = thr16 == 0 || keep_lo(x1, thr16); kp[3] = thr16 == 0 || keep_hi(x1, thr16);
}
template <int NQ>
__device__ __forceinline__ void block_colsum_q(float* lds_row, const f32x4 (&acc)[NQ], int nq, int lane, int wave, int nwave) {
  for (int w = 0; w < nwave; ++w) {          // wave by wave: a fixed summation order
    if (wave == w) {
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        const int q = lane + 64 * c;
        if (q < nq) {
          f32x4 v = acc[c];
          if (w != 0) v += *(const f32x4*)(lds_row + q * 4);
          *(f32x4*)(lds_row + q * 4) = v;
        }
      }
    }
    __syncthreads();
  }
}
__device__ __forceinline__ uint32_t hash_row(const RowDrop& dr, long row_offset, long row) {
  return (uint32_t)(row_offset + (dr.rowmap != nullptr ? (long)dr.rowmap[row] : row));      // row in an SGPR: a scalar load
}

template <int NQ>                  // d == 256 * NQ
__global__ __launch_bounds__(256, (NQ <= 2 ? 6 : NQ == 3 ? 5 : 4)) void add_ln_fwd_q_kernel(const bf16_t* __restrict__ G, const float* __restrict__ X32,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           bf16_t* __restrict__ Y, float* __restrict__ Y32,
                                                           float* __restrict__ mean_o, float* __restrict__ rstd_o, int M, int d,
                                                           float eps, RowDrop dr, long row_offset) {
  const uint32_t key32 = row_key(dr);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nq = d >> 2;
  const float invd = 1.f / (float)d;
  int qo[NQ];                   // element offset of this lane's quads
  f32x4 gm[NQ], bt[NQ];
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    qo[c] = lane * 4 + 256 * c;
    gm[c] = *(const f32x4*)(gamma + qo[c]);
    bt[c] = *(const f32x4*)(beta + qo[c]);
  }
  const long stride = (long)gridDim.x * 4;
  long row = (long)blockIdx.x * 4 + wave;
  if (row >= M) return;
  bf16x4 gc[NQ], gn[NQ];
  f32x4 xc[NQ], xn[NQ];
#pragma unroll
  for (int c = 0; c < NQ; ++c) { gc[c] = *(const bf16x4*)(G + row * d + qo[c]); xc[c] = *(const f32x4*)(X32 + row * d + qo[c]); }
  for (; row < M; row += stride) {
    const long nrow = min(row + stride, (long)M - 1);
#pragma unroll
    for (int c = 0; c < NQ; ++c) {          // the next row: in flight under this row's arithmetic
      gn[c] = *(const bf16x4*)(G + nrow * d + qo[c]);
      xn[c] = *(const f32x4*)(X32 + nrow * d + qo[c]);
    }
    const uint32_t hrow = hash_row(dr, row_offset, row);
    f32x4 s[NQ];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      bool kp[4];
      row_keep4(key32, dr.thr16, hrow, qo[c] >> 2, kp);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gv = kp[j] ? (float)gc[c][j] * dr.inv_keep : 0.f;
        s[c][j] = xc[c][j] + gv;
        sum += s[c][j];
      }
    }
    const float mu = wave_sum(sum) * invd;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NQ; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float t = s[c][j] - mu; sq += t * t; }
    const float rstd = rsqrtf(wave_sum(sq) * invd + eps);
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      f32x4 o;
      bf16x4 ob;
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] = (s[c][j] - mu) * rstd * gm[c][j] + bt[c][j]; ob[j] = (bf16_t)o[j]; }
      *(bf16x4*)(Y + row * d + qo[c]) = ob;
      *(f32x4*)(Y32 + row * d + qo[c]) = o;
    }
    mean_o[row] = mu;             // every lane, one address, one value
    rstd_o[row] = rstd;
#pragma unroll
    for (int c = 0; c < NQ; ++c) { gc[c] = gn[c]; xc[c] = xn[c]; }
  }
}

template <int NQ>
__global__ __launch_bounds__(256, (NQ <= 2 ? 4 : NQ == 3 ? 3 : 2)) void add_ln_bwd_q_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ G,
                                                           const float* __restrict__ X32, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                           bf16_t* __restrict__ dX, bf16_t* __restrict__ dG,
                                                           float* __restrict__ partials, int M, int d, RowDrop dr, long row_offset) {
  const uint32_t key32 = row_key(dr);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;   // [3][d]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nq = d >> 2;
  const float invd = 1.f / (float)d;
  int qo[NQ];
  f32x4 ag[NQ], ab[NQ], abias[NQ], gm[NQ];
#pragma unroll
  for (int c = 0; c < NQ; ++c) {
    qo[c] = lane * 4 + 256 * c;
    ag[c] = ab[c] = abias[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gm[c] = *(const f32x4*)(gamma + qo[c]);
  }
  const long stride = (long)gridDim.x * 4;
  long row = (long)blockIdx.x * 4 + wave;
  if (row < M) {
    bf16x4 gc[NQ], gn[NQ], dc[NQ], dn[NQ];
    f32x4 xc[NQ], xn[NQ];
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      gc[c] = *(const bf16x4*)(G + row * d + qo[c]);
      dc[c] = *(const bf16x4*)(dY + row * d + qo[c]);
      xc[c] = *(const f32x4*)(X32 + row * d + qo[c]);
    }
    for (; row < M; row += stride) {
      const long nrow = min(row + stride, (long)M - 1);
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        gn[c] = *(const bf16x4*)(G + nrow * d + qo[c]);
        dn[c] = *(const bf16x4*)(dY + nrow * d + qo[c]);
        xn[c] = *(const f32x4*)(X32 + nrow * d + qo[c]);
      }
      const float mu = mean_i[row], rstd = rstd_i[row];           // scalar loads (row lives in an SGPR)
      const uint32_t hrow = hash_row(dr, row_offset, row);
      // pass 1: the two row sums and the column accumulators; xhat / dy*gamma are rebuilt in pass 2 from the raw operands
      // instead of living in 24 registers across the reductions (a wave per SIMD more)
      unsigned kbits = 0;
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        bool kp[4];
        row_keep4(key32, dr.thr16, hrow, qo[c] >> 2, kp);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          kbits |= (kp[j] ? 1u : 0u) << (4 * c + j);
          const float gv = kp[j] ? (float)gc[c][j] * dr.inv_keep : 0.f;
          const float xh = (xc[c][j] + gv - mu) * rstd;
          const float dyf = (float)dc[c][j];
          const float dyg = dyf * gm[c][j];
          c1 += dyg;
          c2 += dyg * xh;
          ag[c][j] += dyf * xh;
          ab[c][j] += dyf;
        }
      }
      c1 = wave_sum(c1) * invd;
      c2 = wave_sum(c2) * invd;
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        bf16x4 dsb, dgb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool k = (kbits >> (4 * c + j)) & 1u;
          const float gv = k ? (float)gc[c][j] * dr.inv_keep : 0.f;
          const float xh = (xc[c][j] + gv - mu) * rstd;
          const float dyg = (float)dc[c][j] * gm[c][j];
          const float ds = rstd * (dyg - c1 - xh * c2);
          const float dg = k ? ds * dr.inv_keep : 0.f;
          abias[c][j] += dg;
          dsb[j] = (bf16_t)ds;
          dgb[j] = (bf16_t)dg;
        }
        *(bf16x4*)(dX + row * d + qo[c]) = dsb;
        *(bf16x4*)(dG + row * d + qo[c]) = dgb;
      }
#pragma unroll
      for (int c = 0; c < NQ; ++c) { gc[c] = gn[c]; dc[c] = dn[c]; xc[c] = xn[c]; }
    }
  }
  block_colsum_q<NQ>(red, ag, nq, lane, wave, 4);
  block_colsum_q<NQ>(red + d, ab, nq, lane, wave, 4);
  block_colsum_q<NQ>(red + 2 * d, abias, nq, lane, wave, 4);
  float* out = partials + (long)blockIdx.x * 3 * d;
  for (int t = threadIdx.x; t < 3 * d; t += 256) out[t] = red[t];
}

// Column reduce of per-block partials, fixed order, two-level tree.  partials: [np][nseg*w] fp32.
// block = 32 columns x 8 row-lanes; blockIdx.y = group of `per_group` partial rows.  Pass 1 (np > 64) folds
// groups into a scratch [ng][nseg*w]; the final pass writes segment z of the columns to its own output
// pointer (dgamma / dbeta / dbias live in different .grad buffers), optionally accumulating.
// (The first version -- one thread per column looping over ~1000 partials -- ran on 3 CUs and cost 44 % of
// the step: profiles/r01a_first_run_kernel_stats.csv.)
struct ReduceOut { float* o[3]; };
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ partials, long pstride, int np, int per_group,
                                                        ReduceOut out, long ostride, int w, int nseg, int accumulate) {
  __shared__ float red[8][33];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int seg = blockIdx.z;
  const int col = blockIdx.x * 32 + c;             // column inside the segment
  const int p0 = blockIdx.y * per_group, p1 = min(np, p0 + per_group);
  float s = 0.f;
  if (col < w)
    for (int p = p0 + g; p < p1; p += 8) s += partials[(long)p * pstride + seg * w + col];
  red[g][c] = s;
  __syncthreads();
  if (g == 0 && col < w) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][c];
    // pass 1: one scratch row per group, all segments side by side; final pass: per-segment output
    float* o = ostride != 0 ? out.o[0] + (long)blockIdx.y * ostride + seg * w + col : out.o[seg] + col;
    *o = accumulate ? *o + t : t;
  }
}

// scratch must hold 64 * nseg * w floats when np > 64
static void launch_colreduce(const float* partials, long pstride, int np, ReduceOut out, int w, int nseg, int accumulate,
                             float* scratch, hipStream_t st) {
  const int gx = (w + 31) / 32;
  if (np <= 64 || scratch == nullptr) {
    hipLaunchKernelGGL(colreduce_kernel, dim3(gx, 1, nseg), dim3(256), 0, st, partials, pstride, np, np, out, 0L, w, nseg, accumulate);
    return;
  }
  const int per = (np + 63) / 64;
  const int ng = (np + per - 1) / per;
  ReduceOut tmp; tmp.o[0] = scratch; tmp.o[1] = tmp.o[2] = nullptr;
  hipLaunchKernelGGL(colreduce_kernel, dim3(gx, ng, nseg), dim3(256), 0, st, partials, pstride, np, per, tmp, (long)nseg * w, w, nseg, 0);
  hipLaunchKernelGGL(colreduce_kernel, dim3(gx, 1, nseg), dim3(256), 0, st, scratch, (long)nseg * w, ng, ng, out, 0L, w, nseg, accumulate);
}

// ---- batched second-level column reduce ("launch-boundary reduce") -------------------------------------------
// The per-block partial sums of many producers (LayerNorm dgamma/dbeta/dbias, bias column sums) are reduced by ONE
// launch at the end of backward instead of one or two ~5 us launches behind every producer.  Job j: np partial rows
// of pstride floats hold nseg segments of w columns side by side; out[seg][col] (+)= sum over rows, fixed order.
// Job record = 8 x int64: {partials, pstride, np, w, nseg | accumulate << 8 | first_block << 32, out0, out1, out2}.
__global__ __launch_bounds__(256) void colreduce_batch_kernel(const long long* __restrict__ jobs, int njobs) {
  __shared__ float red[8][33];
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {                                   // wave-uniform binary search on first_block
    const int mid = (lo + hi + 1) >> 1;
    if ((int)(jobs[mid * 8 + 4] >> 32) <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* J = jobs + lo * 8;
  const float* partials = (const float*)J[0];
  const long pstride = J[1];
  const int np = (int)J[2], w = (int)J[3];
  const long long pk = J[4];
  const int accumulate = (int)((pk >> 8) & 0xff);
  const int cb = (int)blockIdx.x - (int)(pk >> 32), gx = (w + 31) >> 5;
  const int seg = cb / gx, col = (cb - seg * gx) * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < w) {
    const float* base = partials + (long)seg * w + col;
    int q = g;
    for (; q + 24 < np; q += 32) {                    // four independent loads in flight per thread
      s0 += base[(long)q * pstride];
      s1 += base[(long)(q + 8) * pstride];
      s2 += base[(long)(q + 16) * pstride];
      s3 += base[(long)(q + 24) * pstride];
    }
    for (; q < np; q += 8) s0 += base[(long)q * pstride];
  }
  red[g][threadIdx.x & 31] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && col < w) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x & 31];
    float* o = (float*)J[5 + seg] + col;
    *o = accumulate ? *o + t : t;
  }
}

// column sums of a bf16 [M,N] matrix (bias grads): per-(slice) partials
__global__ __launch_bounds__(64) void colsum_partial_kernel(const bf16_t* __restrict__ X, long ldx, int M, int N, int rows_per_slice,
                                                            float* __restrict__ partials) {
  const int lane = threadIdx.x;
  const int ch = blockIdx.x * 64 + lane;
  const int slice = blockIdx.y;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (ch * 8 < N) {
    const int r0 = slice * rows_per_slice, r1 = min(M, r0 + rows_per_slice);
    for (int r = r0; r < r1; ++r) {
      float f[8];
      bf8_to_f32(*(const bf16x8*)(X + (long)r * ldx + ch * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) partials[(long)slice * N + ch * 8 + j] = acc[j];
  }
}

// ------------------------------------------------------------------ casts / dropout / expand / rowdot
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  const long nv = n >> 3;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (long)gridDim.x * blockDim.x) {
    const f32x4 a = *(const f32x4*)(src + v * 8), b = *(const f32x4*)(src + v * 8 + 4);
    float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    *(bf16x8*)(dst + v * 8) = f32_to_bf8(f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(nv << 3) + threadIdx.x] = (bf16_t)src[(nv << 3) + threadIdx.x];
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n) {
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (long)gridDim.x * blockDim.x) dst[v] = (float)src[v];
}

__global__ void dropout_kernel(const bf16_t* __restrict__ X, bf16_t* __restrict__ Y, long M, int N, RowDrop dr, long row_offset) {
  const uint32_t key32 = row_key(dr);
  const int nch = N >> 3;
  const long nv = M * nch;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (long)gridDim.x * blockDim.x) {
    const long row = v / nch;
    const int ch = (int)(v - row * nch);
    float f[8];
    bf8_to_f32(*(const bf16x8*)(X + v * 8), f);
    bool kp8[8];
    row_keep8(key32, dr.thr16, (uint32_t)(row_offset + (dr.rowmap != nullptr ? (long)dr.rowmap[row] : (long)row)), ch, kp8);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = kp8[j] ? f[j] * dr.inv_keep : 0.f;
    *(bf16x8*)(Y + v * 8) = f32_to_bf8(f);
  }
}

// out[b][e][:] = q[e][:]   (emotion_decoder.py:127)
__global__ void expand_rows_kernel(const float* __restrict__ q, bf16_t* __restrict__ out, float* __restrict__ out32, int B, long n) {
  const long total = (long)B * n;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (long)gridDim.x * blockDim.x) {
    const float x = q[v % n];
    out[v] = (bf16_t)x;
    if (out32 != nullptr) out32[v] = x;       // fp32 twin of the residual stream (a torch expand + copy launch before)
  }
}
// dropout seed word of a captured step: += the 64-bit golden ratio, once per replay (a torch add launch before)
__global__ void seed_bump_kernel(unsigned long long* seed) { if (threadIdx.x == 0) *seed += 0x9E3779B97F4A7C15ull; }

// logits[r] = z[r,:] . w + b   (emotion_decoder.py:155)
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const bf16_t* __restrict__ Z, const float* __restrict__ Z32, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ out, int M, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  float s = 0.f;
  for (int ch = lane; ch < (d >> 3); ch += 64) {
    float f[8];
    load_resid(Z, Z32, (long)row * d + ch * 8, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j] * w[ch * 8 + j];
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s + (b != nullptr ? b[0] : 0.f);
}
// dZ[r,:] = dl[r]*w ; dw[e] = sum_r dl[r] z[r,e] ; db = sum_r dl[r]   (M is small: B*N_e)
// 32 columns per block and ROWDOT_GROUPS row groups of 32 lanes: the kernel sits on the step's serial chain right behind the loss
// and its duration is the length of a group's row loop (8 groups x 48 rows at cfg 2: 17 us; 32 groups x 12 rows: see HISTORY)
constexpr int ROWDOT_GROUPS = 32;
__global__ __launch_bounds__(ROWDOT_GROUPS * 32) void rowdot_bwd_kernel(const float* __restrict__ dl, const bf16_t* __restrict__ Z, const float* __restrict__ Z32, const float* __restrict__ w,
                                                         bf16_t* __restrict__ dZ, float* __restrict__ dw, float* __restrict__ db, int M, int d,
                                                         int accumulate) {
  __shared__ float red[ROWDOT_GROUPS][33];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c;
  float acc = 0.f, sb = 0.f;
  if (col < d) {
    const float wc = w[col];
    for (int r = g; r < M; r += ROWDOT_GROUPS) {
      const float gr = dl[r];
      acc += gr * (Z32 != nullptr ? Z32[(long)r * d + col] : (float)Z[(long)r * d + col]);
      dZ[(long)r * d + col] = (bf16_t)(gr * wc);
      sb += gr;
    }
  }
  red[g][c] = acc;
  __syncthreads();
  if (g == 0 && col < d) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < ROWDOT_GROUPS; ++k) t += red[k][c];
    dw[col] = accumulate ? dw[col] + t : t;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    red[g][c] = (c == 0) ? sb : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int k = 0; k < ROWDOT_GROUPS; ++k) t += red[k][0];
      db[0] = accumulate ? db[0] + t : t;
    }
  }
}

// ------------------------------------------------------------------ beta gate
// LayerNorm every row of X[b, :, :]; write the first Lkeep rows; pooled partial sums over valid rows.
template <int NCH, bool PF>
__device__ __forceinline__ void ln_pool_fwd_body(const bf16_t* __restrict__ X, const float* __restrict__ X32, const uint8_t* __restrict__ mask,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 bf16_t* __restrict__ Yn, float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                 float* __restrict__ partials, int L, int Lkeep, int d, float eps, int chunk, int nchunks) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, nchunk = d >> 3;
  const float invd = 1.f / (float)d;
  float acc[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
  constexpr bool HOIST = NCH <= 4;
  float gm8[HOIST ? NCH : 1][8], bt8[HOIST ? NCH : 1][8];
  if (HOIST) { load_cols<HOIST ? NCH : 1>(gamma, d >> 3, threadIdx.x & 63, gm8); load_cols<HOIST ? NCH : 1>(beta, d >> 3, threadIdx.x & 63, bt8); }
  // a wave walks its 8 rows one after the other and every row is one trip to HBM: the NEXT row's loads are issued before this
  // row's reductions, so the trips overlap (the kernel sat at 2 TB/s with one row in flight per wave)
  const int lend = min(L, chunk * 32 + 32);
  float nx[NCH][8];
  auto load_row = [&](int l, float (&dst)[NCH][8]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) load_resid(X, X32, ((long)b * L + l) * d + ch * 8, dst[c]);
    }
  };
  if (PF && chunk * 32 + wave < lend) load_row(chunk * 32 + wave, nx);
  for (int l = chunk * 32 + wave; l < lend; l += 4) {
    const long row = (long)b * L + l;
    float s[NCH][8];
    float sum = 0.f;
    if (!PF) load_row(l, nx);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[c][j] = nx[c][j]; sum += s[c][j]; }
      }
    }
    if (PF && l + 4 < lend) load_row(l + 4, nx);
    const float mu = wave_sum(sum) * invd;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (lane + 64 * c < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = s[c][j] - mu; sq += t * t; }
      }
    const float rstd = rsqrtf(wave_sum(sq) * invd + eps);
    const bool valid = mask == nullptr || mask[(long)b * L + l] == 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o[j] = (s[c][j] - mu) * rstd * (HOIST ? gm8[HOIST ? c : 0][j] : gamma[ch * 8 + j]) + (HOIST ? bt8[HOIST ? c : 0][j] : beta[ch * 8 + j]);
          if (valid) acc[c][j] += o[j];
        }
        if (l < Lkeep) *(bf16x8*)(Yn + ((long)b * Lkeep + l) * d + ch * 8) = f32_to_bf8(o);
      }
    }
    if (lane == 0) { mean_o[row] = mu; rstd_o[row] = rstd; }
  }
  block_colsum<NCH>(red, acc, nchunk, lane, wave, 4);
  float* out = partials + ((long)b * nchunks + chunk) * d;
  for (int t = threadIdx.x; t < d; t += 256) out[t] = red[t];
}
template <int NCH, bool PF>
__global__ __launch_bounds__(256) void ln_pool_fwd_kernel(const bf16_t* __restrict__ X, const float* __restrict__ X32, const uint8_t* __restrict__ mask,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          bf16_t* __restrict__ Yn, float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                          float* __restrict__ partials, int L, int Lkeep, int d, float eps) {
  ln_pool_fwd_body<NCH, PF>(X, X32, mask, gamma, beta, Yn, mean_o, rstd_o, partials, L, Lkeep, d, eps, blockIdx.x, gridDim.x);
}
// Both modalities of the gate from ONE launch: blocks [0, nc of side 0) walk side 0, the rest side 1.  On two streams the pair
// cost a fork and a join around each of the gate's two LayerNorm + pool steps (10-27 us of idle device each, profiles/
// r04_step_timeline.txt); the blocks are the single kernel's, so are the results.
struct PoolFwdSide { const bf16_t* X; const float* X32; const uint8_t* mask; const float* gamma; const float* beta; bf16_t* Yn; float* mean; float* rstd; float* partials; int L, Lkeep, nc; };
struct PoolFwdPair { PoolFwdSide s[2]; };
#define PAIR_PICK(f) (second ? p.s[1].f : p.s[0].f)      // field by field: an indexed copy of the argument struct would go to scratch
template <int NCH, bool PF>
__global__ __launch_bounds__(256, 4) void ln_pool_fwd_pair_kernel(const PoolFwdPair p, int d, float eps) {
  const bool second = (int)blockIdx.x >= p.s[0].nc;
  ln_pool_fwd_body<NCH, PF>(PAIR_PICK(X), PAIR_PICK(X32), PAIR_PICK(mask), PAIR_PICK(gamma), PAIR_PICK(beta), PAIR_PICK(Yn), PAIR_PICK(mean),
                            PAIR_PICK(rstd), PAIR_PICK(partials), PAIR_PICK(L), PAIR_PICK(Lkeep), d, eps,
                            (int)blockIdx.x - (second ? p.s[0].nc : 0), PAIR_PICK(nc));
}

// pooled means + gate input [a, t, |a-t|, a*t]  (beta_gate_tacfn.py:83-89)
__global__ __launch_bounds__(256) void gate_input_kernel(const float* __restrict__ pa, int nca, const float* __restrict__ pt, int nct,
                                                         const uint8_t* __restrict__ mask_a, const uint8_t* __restrict__ mask_t, int La, int Lt,
                                                         int d, bf16_t* __restrict__ gin, float* __restrict__ a_pool, float* __restrict__ t_pool,
                                                         float* __restrict__ cnt) {
  // on the step's serial chain: every load of a phase is requested before the first one is used.  The valid counts are small
  // integers (any summation order gives the same float); the partial sums are added in chunk order, as before.
  __shared__ float sc[2], wc[4][2];
  const int b = blockIdx.x, tid = threadIdx.x;
  float ca = 0.f, ct = 0.f;
  for (int l = tid; l < La; l += 256) ca += (mask_a == nullptr || mask_a[(long)b * La + l] == 0) ? 1.f : 0.f;
  for (int l = tid; l < Lt; l += 256) ct += (mask_t == nullptr || mask_t[(long)b * Lt + l] == 0) ? 1.f : 0.f;
  ca = wave_sum(ca); ct = wave_sum(ct);
  if ((tid & 63) == 0) { wc[tid >> 6][0] = ca; wc[tid >> 6][1] = ct; }
  __syncthreads();
  if (tid == 0) {
    sc[0] = fmaxf(wc[0][0] + wc[1][0] + wc[2][0] + wc[3][0], 1.f); sc[1] = fmaxf(wc[0][1] + wc[1][1] + wc[2][1] + wc[3][1], 1.f);
    cnt[b * 2] = sc[0]; cnt[b * 2 + 1] = sc[1];
  }
  __syncthreads();
  const float ia = 1.f / sc[0], it = 1.f / sc[1];
  for (int c = tid; c < d; c += 256) {
    float a = 0.f, t = 0.f;
    for (int k0 = 0; k0 < nca; k0 += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = k0 + k < nca ? pa[((long)b * nca + k0 + k) * d + c] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) if (k0 + k < nca) a += v[k];
    }
    for (int k0 = 0; k0 < nct; k0 += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = k0 + k < nct ? pt[((long)b * nct + k0 + k) * d + c] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) if (k0 + k < nct) t += v[k];
    }
    a *= ia; t *= it;
    a_pool[(long)b * d + c] = a; t_pool[(long)b * d + c] = t;
    bf16_t* g = gin + (long)b * 4 * d;
    g[c] = (bf16_t)a; g[d + c] = (bf16_t)t; g[2 * d + c] = (bf16_t)fabsf(a - t); g[3 * d + c] = (bf16_t)(a * t);
  }
}

// w = sigmoid(pre), beta = mean_d(w)   (beta_gate_tacfn.py:92-95)
__global__ __launch_bounds__(256) void sigmoid_beta_kernel(const float* __restrict__ pre, float* __restrict__ w, float* __restrict__ beta, int d) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int c = tid; c < d; c += 256) {
    const float v = 1.f / (1.f + __expf(-pre[(long)b * d + c]));
    w[(long)b * d + c] = v;
    s += v;
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) beta[b] = (red[0] + red[1] + red[2] + red[3]) / (float)d;
}

// h = w*a + (1-w)*t over [B, L, d]   (beta_gate_tacfn.py:113-116)
__global__ void fuse_fwd_kernel(const float* __restrict__ w, const bf16_t* __restrict__ A, const bf16_t* __restrict__ T,
                                bf16_t* __restrict__ H, int B, int L, int d) {
  const int nch = d >> 3;
  const long nv = (long)B * L * nch;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(v % nch);
    const long b = v / ((long)L * nch);
    float a[8], t[8], o[8];
    bf8_to_f32(*(const bf16x8*)(A + v * 8), a);
    bf8_to_f32(*(const bf16x8*)(T + v * 8), t);
    const float* wp = w + b * d + ch * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = wp[j] * a[j] + (1.f - wp[j]) * t[j];
    *(bf16x8*)(H + v * 8) = f32_to_bf8(o);
  }
}

// dw partials: sum_l dH[b,l,:] * (A - T)[b,l,:]
template <int NCH>
__global__ __launch_bounds__(256) void fuse_bwd_dw_kernel(const bf16_t* __restrict__ dH, const bf16_t* __restrict__ A, const bf16_t* __restrict__ T,
                                                          float* __restrict__ partials, int L, int d) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = d >> 3;
  float acc[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
  for (int l = chunk * 32 + wave; l < min(L, chunk * 32 + 32); l += 4) {
    const long row = (long)b * L + l;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        float g[8], a[8], t[8];
        bf8_to_f32(*(const bf16x8*)(dH + row * d + ch * 8), g);
        bf8_to_f32(*(const bf16x8*)(A + row * d + ch * 8), a);
        bf8_to_f32(*(const bf16x8*)(T + row * d + ch * 8), t);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[c][j] += g[j] * (a[j] - t[j]);
      }
    }
  }
  block_colsum<NCH>(red, acc, nchunk, lane, wave, 4);
  float* out = partials + ((long)b * gridDim.x + chunk) * d;
  for (int t = threadIdx.x; t < d; t += 256) out[t] = red[t];
}

// dpre = (sum partials + dbeta/d) * w * (1-w)  -> bf16 for the gate-MLP backward GEMMs
__global__ __launch_bounds__(256) void gate_dpre_kernel(const float* __restrict__ partials, int np, const float* __restrict__ dbeta,
                                                        const float* __restrict__ w, bf16_t* __restrict__ dpre, int d) {
  const int b = blockIdx.x;
  const float db = dbeta != nullptr ? dbeta[b] / (float)d : 0.f;
  for (int c = threadIdx.x; c < d; c += 256) {
    float s = db;
    const float wv = w[(long)b * d + c];
    for (int k0 = 0; k0 < np; k0 += 8) {        // loads of a chunk requested together, added in chunk order
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = k0 + k < np ? partials[((long)b * np + k0 + k) * d + c] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) if (k0 + k < np) s += v[k];
    }
    dpre[(long)b * d + c] = (bf16_t)(s * wv * (1.f - wv));
  }
}

// d gate_in [B,4d] -> d a_pool, d t_pool (already divided by the valid counts)
__global__ __launch_bounds__(256) void gate_input_bwd_kernel(const bf16_t* __restrict__ dgin, const float* __restrict__ a_pool,
                                                             const float* __restrict__ t_pool, const float* __restrict__ cnt,
                                                             float* __restrict__ da, float* __restrict__ dt, int d) {
  const int b = blockIdx.x;
  const float ia = 1.f / cnt[b * 2], it = 1.f / cnt[b * 2 + 1];
  const bf16_t* g = dgin + (long)b * 4 * d;
  for (int c = threadIdx.x; c < d; c += 256) {
    const float a = a_pool[(long)b * d + c], t = t_pool[(long)b * d + c];
    const float g0 = (float)g[c], g1 = (float)g[d + c], g2 = (float)g[2 * d + c], g3 = (float)g[3 * d + c];
    const float sg = (a > t) ? 1.f : ((a < t) ? -1.f : 0.f);
    da[(long)b * d + c] = (g0 + sg * g2 + t * g3) * ia;
    dt[(long)b * d + c] = (g1 - sg * g2 + a * g3) * it;
  }
}

// rows of one sample that one block of ln_pool_bwd walks (4 waves, POOL_BWD_ROWS / 4 rows each); its partial sums are per block
constexpr int POOL_BWD_ROWS = 16;

// backward of LayerNorm+pool+fuse-branch for one modality:
//   dYn[l] = (l < Lf ? coef * dH[b,l] : 0) + (valid_l ? dpool[b] : 0),  coef = is_a ? w : 1-w ;  dX = LN'(dYn)
template <int NCH>
__device__ __forceinline__ void ln_pool_bwd_body(const bf16_t* __restrict__ dH, int Lf, const float* __restrict__ w, int is_a,
                                                 const float* __restrict__ dpool, const uint8_t* __restrict__ mask,
                                                 const bf16_t* __restrict__ X, const float* __restrict__ X32, const float* __restrict__ gamma,
                                                 const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                 bf16_t* __restrict__ dX, float* __restrict__ partials, int L, int d, int chunk, int nchunks) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = (float*)smem_raw;   // [2][d] column-sum scratch (+ [3][d] constants for d <= 1024)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, nchunk = d >> 3;
  const float invd = 1.f / (float)d;
  float ag[NCH][8], ab[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) { ag[c][j] = 0.f; ab[c][j] = 0.f; }
  // per-column constants of this sample (pooled gradient, gate weight, LN gain), loaded once per block instead of once per row
  // (as scalar loads inside the row loop they made this kernel 3x slower than its traffic): in registers for wide rows; for
  // d <= 1024 in LDS behind the column-sum scratch (48 registers less: three waves per SIMD instead of two beside the prefetch)
  constexpr bool CLDS = NCH <= 2;
  float* cst = red + 2 * d;   // [3][d], CLDS only
  float dp8[CLDS ? 1 : NCH][8], wv8[CLDS ? 1 : NCH][8], gm8[CLDS ? 1 : NCH][8];
  if (CLDS) {
    for (int t = threadIdx.x; t < d; t += 256) {
      const float wt = dH != nullptr ? w[(long)b * d + t] : 0.f;
      cst[t] = dpool[(long)b * d + t];
      cst[d + t] = is_a ? wt : 1.f - wt;
      cst[2 * d + t] = gamma[t];
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int c = 0; c < (CLDS ? 1 : NCH); ++c) {
      const int ch = lane + 64 * c;
#pragma unroll
      for (int j = 0; j < 8; ++j) { dp8[c][j] = 0.f; wv8[c][j] = 0.f; gm8[c][j] = 0.f; }
      if (ch < nchunk) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 a = *(const f32x4*)(dpool + (long)b * d + ch * 8 + h * 4);
          const f32x4 g4 = *(const f32x4*)(gamma + ch * 8 + h * 4);
          f32x4 w4 = {0.f, 0.f, 0.f, 0.f};
          if (dH != nullptr) w4 = *(const f32x4*)(w + (long)b * d + ch * 8 + h * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dp8[c][h * 4 + e] = a[e];
            gm8[c][h * 4 + e] = g4[e];
            wv8[c][h * 4 + e] = is_a ? w4[e] : 1.f - w4[e];
          }
        }
      }
    }
  }
  // a wave walks its POOL_BWD_ROWS / 4 rows one after the other and every row is one trip to HBM: the next row's operands are
  // requested before this row's reductions (one row in flight per wave and 178 registers, i.e. two waves per SIMD, held the
  // kernel at 2.5 TB/s; 32-row chunks also left a second, one-third-full round of blocks on the 256 CUs)
  const int lbeg = chunk * POOL_BWD_ROWS + wave, lend = min(L, chunk * POOL_BWD_ROWS + POOL_BWD_ROWS);
  const bool has_dh = dH != nullptr;
  float nxx[NCH][8];
  bf16x8 nxh[NCH];
  auto load_row = [&](int l) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        load_resid(X, X32, ((long)b * L + l) * d + ch * 8, nxx[c]);
        if (has_dh && l < Lf) nxh[c] = *(const bf16x8*)(dH + ((long)b * Lf + l) * d + ch * 8);
      }
    }
  };
  constexpr bool PF = NCH <= 2;   // wider rows (d > 1024) would spill with a second row in registers
  if (PF) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int j = 0; j < 8; ++j) nxx[c][j] = 0.f;
      nxh[c] = bf16x8{};
    }
  }
  if (PF && lbeg < lend) load_row(lbeg);
  for (int l = lbeg; l < lend; l += 4) {
    const long row = (long)b * L + l;
    const float mu = mean_i[row], rstd = rstd_i[row];
    const bool valid = mask == nullptr || mask[row] == 0;
    const bool grad_row = has_dh && l < Lf;
    float xh[NCH][8], dyg[NCH][8];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        float xf[8], gh[8], dpc[8], wvc[8], gmc[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 a, wq, g4;
          if (CLDS) {
            a = *(const f32x4*)(cst + ch * 8 + h * 4); wq = *(const f32x4*)(cst + d + ch * 8 + h * 4); g4 = *(const f32x4*)(cst + 2 * d + ch * 8 + h * 4);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dpc[h * 4 + e] = CLDS ? a[e] : dp8[CLDS ? 0 : c][h * 4 + e];
            wvc[h * 4 + e] = CLDS ? wq[e] : wv8[CLDS ? 0 : c][h * 4 + e];
            gmc[h * 4 + e] = CLDS ? g4[e] : gm8[CLDS ? 0 : c][h * 4 + e];
          }
        }
        if (PF) {
#pragma unroll
          for (int j = 0; j < 8; ++j) xf[j] = nxx[c][j];
          bf8_to_f32(nxh[c], gh);
        } else {
          load_resid(X, X32, row * d + ch * 8, xf);
          if (grad_row) bf8_to_f32(*(const bf16x8*)(dH + ((long)b * Lf + l) * d + ch * 8), gh);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float dy = valid ? dpc[j] : 0.f;
          if (grad_row) dy += wvc[j] * gh[j];
          xh[c][j] = (xf[j] - mu) * rstd;
          dyg[c][j] = dy * gmc[j];
          c1 += dyg[c][j];
          // keeps the SLP vectoriser from pairing steps of this serial sum: with the prefetch in the loop it chose the packed add
          // whose low half reads the high dword of src1, the form `make check-isa` rejects (Makefile, ATTN_FLAGS)
          if (PF) asm volatile("" : "+v"(c1));
          c2 += dyg[c][j] * xh[c][j];
          ag[c][j] += dy * xh[c][j];
          ab[c][j] += dy;
        }
      }
    }
    if (PF && l + 4 < lend) load_row(l + 4);
    c1 = wave_sum(c1) * invd;
    c2 = wave_sum(c2) * invd;
    asm volatile("" : "+v"(c1));      // two separate registers: as a pair they were fed to packed FMAs through the operand
    asm volatile("" : "+v"(c2));      // select `make check-isa` rejects (wide rows, once this body was shared by two kernels)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        float ds[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ds[j] = rstd * (dyg[c][j] - c1 - xh[c][j] * c2);
        *(bf16x8*)(dX + row * d + ch * 8) = f32_to_bf8(ds);
      }
    }
  }
  block_colsum<NCH>(red, ag, nchunk, lane, wave, 4);
  block_colsum<NCH>(red + d, ab, nchunk, lane, wave, 4);
  float* out = partials + ((long)b * nchunks + chunk) * 2 * d;
  for (int t = threadIdx.x; t < 2 * d; t += 256) out[t] = red[t];
}
template <int NCH>
__global__ __launch_bounds__(256) void ln_pool_bwd_kernel(const bf16_t* __restrict__ dH, int Lf, const float* __restrict__ w, int is_a,
                                                          const float* __restrict__ dpool, const uint8_t* __restrict__ mask,
                                                          const bf16_t* __restrict__ X, const float* __restrict__ X32, const float* __restrict__ gamma,
                                                          const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                          bf16_t* __restrict__ dX, float* __restrict__ partials, int L, int d) {
  ln_pool_bwd_body<NCH>(dH, Lf, w, is_a, dpool, mask, X, X32, gamma, mean_i, rstd_i, dX, partials, L, d, blockIdx.x, gridDim.x);
}
// both modalities from one launch (see ln_pool_fwd_pair_kernel): side 0 = audio (coefficient w), side 1 = text (1 - w)
struct PoolBwdSide { const float* dpool; const uint8_t* mask; const bf16_t* X; const float* X32; const float* gamma; const float* mean; const float* rstd; bf16_t* dX; float* partials; int L, nc; };
struct PoolBwdPair { PoolBwdSide s[2]; };
template <int NCH>
__global__ __launch_bounds__(256) void ln_pool_bwd_pair_kernel(const bf16_t* __restrict__ dH, int Lf, const float* __restrict__ w, const PoolBwdPair p, int d) {
  const bool second = (int)blockIdx.x >= p.s[0].nc;
  ln_pool_bwd_body<NCH>(dH, Lf, w, second ? 0 : 1, PAIR_PICK(dpool), PAIR_PICK(mask), PAIR_PICK(X), PAIR_PICK(X32), PAIR_PICK(gamma), PAIR_PICK(mean),
                        PAIR_PICK(rstd), PAIR_PICK(dX), PAIR_PICK(partials), PAIR_PICK(L), d, (int)blockIdx.x - (second ? p.s[0].nc : 0), PAIR_PICK(nc));
}

// ================================================================== host entry points
#define DISPATCH_NCH(d, CALL)                         \
  {                                                   \
    const int nch__ = ((d) / 8 + 63) / 64;            \
    if (nch__ <= 1) { CALL(1); }                      \
    else if (nch__ <= 2) { CALL(2); }                 \
    else if (nch__ <= 4) { CALL(4); }                 \
    else { CALL(8); }                                 \
  }

// Which add_ln kernels run: 1 (default) the chunk-mapped ones, 0 the quad-mapped, software-pipelined ones of round 4 (d = 256 k).
// Measured alone on the cfg-2 shapes (scripts_dev/bench_rows.py, profiles/r04_rows.log): forward 38.6 -> 36.8 us (audio rows),
// 16.0 -> 14.8 us (text rows), backward 19.1 -> 17.7 us (text), audio backward unchanged -- both mappings already move 5.5-6.4
// TB/s alone, i.e. what HBM3E gives; ~0.03 ms of an 8 ms step.  The two mappings draw the same dropout masks and differ by the
// summation order of the row statistics only, but that is enough to change a bf16 rounding of y in a few elements per
// thousand, and on the ill-conditioned closed-form fixtures such a perturbation moves the per-parameter gradient errors by
// 2-4 % of themselves (both mappings are equally far from the fp32 oracle: scripts_dev/diag_grads_variant2.py) -- enough to
// push two parameters over bounds that were calibrated on the chunk mapping's draw.  The chunk mapping therefore stays the
// numerics of record; the quad kernels are built, tested against it (tests/test_gpu_kernels.py) and selectable here.
static int g_rowops_variant = 1;
extern "C" int hriemo_rowops_force_variant(int v) {
  g_rowops_variant = v == 0 ? 0 : 1;
  return 0;
}
static bool use_quad(int d) { return g_rowops_variant == 0 && d <= 1024 && d % 256 == 0; }
#define DISPATCH_NQ(d, CALL)                  \
  {                                           \
    const int nq__ = (d) / 256;               \
    if (nq__ <= 1) { CALL(1); }               \
    else if (nq__ == 2) { CALL(2); }          \
    else if (nq__ == 3) { CALL(3); }          \
    else { CALL(4); }                         \
  }

static int check_rows(int M, int d) {
  HRIEMO_CHECK(M > 0 && d > 0, "rowops: empty problem");
  HRIEMO_CHECK(d % 8 == 0 && d <= 4096, "rowops: d=%d must be a multiple of 8 and <= 4096", d);
  return 0;
}
static int row_grid(int M, int cap) { int g = (M + 3) / 4; return g > cap ? cap : g; }
// add_ln_bwd blocks are long-lived (grid-stride over rows, column partials at the end): the grid is exactly the number
// of blocks the chip holds at once (occupancy x CUs, 3 x 256 for d = 768).  More blocks than that run as a second,
// mostly empty round: 1024 blocks took 60.6 us on 25600 x 768, 768 take 50.4.
static int num_cus_rowops() {
  int dev = 0, cus = 256;
  hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  return cus;
}
static int lnb_cap(int d) {
  static int cache[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};      // chunk-mapped by NCH; quad-mapped by NQ
  const bool quad = use_quad(d);
  const int nch = (d / 8 + 63) / 64;
  const int slot = quad ? d / 256 - 1 : (nch <= 1 ? 0 : nch <= 2 ? 1 : nch <= 4 ? 2 : 3);
  int& c = cache[quad ? 1 : 0][slot];
  if (c == 0) {
    int per = 0;
    if (quad) {
#define CALL(N) hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)add_ln_bwd_q_kernel<N>, 256, (size_t)3 * d * 4)
      DISPATCH_NQ(d, CALL)
#undef CALL
    } else {
#define CALL(N) hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)add_ln_bwd_kernel<N>, 256, (size_t)3 * d * 4)
      DISPATCH_NCH(d, CALL)
#undef CALL
    }
    if (per < 1) per = 3;
    c = per * num_cus_rowops();
  }
  return c;
}
// the quad-mapped forward is persistent as well: one block per resident slot, rows walked with a grid stride
static int lnf_cap(int d) {
  static int cache[4] = {0, 0, 0, 0};
  int& c = cache[d / 256 - 1];
  if (c == 0) {
    int per = 0;
#define CALL(N) hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)add_ln_fwd_q_kernel<N>, 256, 0)
    DISPATCH_NQ(d, CALL)
#undef CALL
    if (per < 1) per = 4;
    c = per * num_cus_rowops();
  }
  return c;
}

static int add_ln_fwd_impl(const void* G, const void* X, const float* X32, const float* gamma, const float* beta, void* Y,
                           float* Y32, float* mean, float* rstd, int M, int d, float eps, float p_drop, unsigned long long seed,
                           const unsigned long long* seed_dev, unsigned site, long row_offset, void* Yq, void* SY, long ldsy,
                           const long long* row_index, hipStream_t st) {
  if (check_rows(M, d)) return 1;
  HRIEMO_CHECK(Yq == nullptr || (d % 32 == 0 && SY != nullptr && ldsy >= M), "add_ln_fwd: the MX-fp8 copy needs d %% 32 == 0 and a scale buffer");
  RowDrop dr = row_drop(p_drop, seed, seed_dev, site);
  dr.rowmap = row_index;
  hriemo_prof_begin(HP_ROWOPS, st);
  if (use_quad(d) && Yq == nullptr && X32 != nullptr && Y32 != nullptr) {        // the model's form: fp32 twin in and out
#define CALL(N) hipLaunchKernelGGL((add_ln_fwd_q_kernel<N>), dim3(row_grid(M, lnf_cap(d))), dim3(256), 0, st, (const bf16_t*)G, X32, gamma, beta, (bf16_t*)Y, Y32, mean, rstd, M, d, eps, dr, row_offset)
    DISPATCH_NQ(d, CALL)
#undef CALL
  } else {
#define CALL(N) hipLaunchKernelGGL((add_ln_fwd_kernel<N>), dim3(row_grid(M, 4096)), dim3(256), 0, st, (const bf16_t*)G, (const bf16_t*)X, X32, gamma, beta, (bf16_t*)Y, Y32, mean, rstd, M, d, eps, dr, row_offset, (uint8_t*)Yq, (uint8_t*)SY, ldsy)
    DISPATCH_NCH(d, CALL)
#undef CALL
  }
  HRIEMO_LAUNCH_CHECK("add_ln_fwd_kernel");
  hriemo_prof_end(HP_ROWOPS, st, ((X32 ? 3.0 : (X ? 2.0 : 1.0)) + 1.0 + (Y32 ? 2.0 : 0.0) + (Yq ? 0.5 : 0.0)) * M * d * 2);
  return 0;
}
extern "C" int hriemo_add_ln_fwd(const void* G, const void* X, const float* X32, const float* gamma, const float* beta, void* Y,
                                 float* Y32, float* mean, float* rstd, int M, int d, float eps, float p_drop, unsigned long long seed,
                                 const unsigned long long* seed_dev, unsigned site, long row_offset, hipStream_t st) {
  return add_ln_fwd_impl(G, X, X32, gamma, beta, Y, Y32, mean, rstd, M, d, eps, p_drop, seed, seed_dev, site, row_offset, nullptr, nullptr, 0, nullptr, st);
}
// same, plus the MX-fp8 copy of Y (bytes [M][d], scales [d/32][ldsy]) for the next projection / FFN GEMM (hriemo_gemm_mx8)
extern "C" int hriemo_add_ln_fwd_mx8(const void* G, const void* X, const float* X32, const float* gamma, const float* beta, void* Y,
                                     float* Y32, float* mean, float* rstd, int M, int d, float eps, float p_drop, unsigned long long seed,
                                     const unsigned long long* seed_dev, unsigned site, long row_offset, void* Yq, void* SY, long ldsy,
                                     hipStream_t st) {
  HRIEMO_CHECK(Yq != nullptr, "add_ln_fwd_mx8: Yq required");
  return add_ln_fwd_impl(G, X, X32, gamma, beta, Y, Y32, mean, rstd, M, d, eps, p_drop, seed, seed_dev, site, row_offset, Yq, SY, ldsy, nullptr, st);
}

// same as hriemo_add_ln_fwd (Yq / SY may be NULL) with the dropout hash keyed by row_index[row] instead of row
extern "C" int hriemo_add_ln_fwd_rows(const void* G, const void* X, const float* X32, const float* gamma, const float* beta, void* Y,
                                      float* Y32, float* mean, float* rstd, int M, int d, float eps, float p_drop, unsigned long long seed,
                                      const unsigned long long* seed_dev, unsigned site, long row_offset, void* Yq, void* SY, long ldsy,
                                      const long long* row_index, hipStream_t st) {
  return add_ln_fwd_impl(G, X, X32, gamma, beta, Y, Y32, mean, rstd, M, d, eps, p_drop, seed, seed_dev, site, row_offset, Yq, SY, ldsy,
                         row_index, st);
}

extern "C" long hriemo_add_ln_bwd_workspace_bytes(int M, int d) { return ((long)row_grid(M, lnb_cap(d)) * 3 * d + 64L * 3 * d) * 4; }

static int add_ln_bwd_impl(const void* dY, const void* G, const void* X, const float* X32, const float* gamma, const float* mean,
                           const float* rstd, void* dX, void* dG, float* dgamma, float* dbeta, float* dbias, int accumulate,
                           int M, int d, float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                           unsigned site, long row_offset, float* workspace, const long long* row_index, hipStream_t st) {
  if (check_rows(M, d)) return 1;
  HRIEMO_CHECK(workspace != nullptr, "add_ln_bwd: workspace required");
  RowDrop dr = row_drop(p_drop, seed, seed_dev, site);
  dr.rowmap = row_index;
  const int nb = row_grid(M, lnb_cap(d));
  hriemo_prof_begin(HP_ROWOPS, st);
  if (dG == nullptr && p_drop == 0.f && dX != nullptr && use_quad(d) && X32 != nullptr) dG = dX;      // no dropout: dG == dX, one tensor, equal bytes twice
  if (use_quad(d) && X32 != nullptr && dX != nullptr && dG != nullptr) {
#define CALL(N) hipLaunchKernelGGL((add_ln_bwd_q_kernel<N>), dim3(nb), dim3(256), 3 * d * 4, st, (const bf16_t*)dY, (const bf16_t*)G, X32, gamma, mean, rstd, (bf16_t*)dX, (bf16_t*)dG, workspace, M, d, dr, row_offset)
    DISPATCH_NQ(d, CALL)
#undef CALL
  } else {
#define CALL(N) hipLaunchKernelGGL((add_ln_bwd_kernel<N>), dim3(nb), dim3(256), 3 * d * 4, st, (const bf16_t*)dY, (const bf16_t*)G, (const bf16_t*)X, X32, gamma, mean, rstd, (bf16_t*)dX, (bf16_t*)dG, workspace, M, d, dr, row_offset)
    DISPATCH_NCH(d, CALL)
#undef CALL
  }
  HRIEMO_LAUNCH_CHECK("add_ln_bwd_kernel");
  // bytes moved: dY, G, dX, dG as bf16 + the residual operand (fp32 twin: 4 bytes per element)
  hriemo_prof_end(HP_ROWOPS, st, ((dX ? 1.0 : 0.0) + (dG ? 1.0 : 0.0) + 2.0 + (X32 ? 2.0 : (X ? 1.0 : 0.0))) * M * d * 2);
  if (dgamma == nullptr) return 0;      // caller reduces the [nb][3d] partials later (hriemo_colreduce_batch)
  float* scratch = workspace + (long)nb * 3 * d;
  ReduceOut ro; ro.o[0] = dgamma; ro.o[1] = dbeta; ro.o[2] = dbias;
  launch_colreduce(workspace, (long)3 * d, nb, ro, d, dbias != nullptr ? 3 : 2, accumulate, scratch, st);
  HRIEMO_LAUNCH_CHECK("colreduce_kernel");
  return 0;
}

extern "C" int hriemo_add_ln_bwd(const void* dY, const void* G, const void* X, const float* X32, const float* gamma, const float* mean,
                                 const float* rstd, void* dX, void* dG, float* dgamma, float* dbeta, float* dbias, int accumulate,
                                 int M, int d, float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                                 unsigned site, long row_offset, float* workspace, hipStream_t st) {
  return add_ln_bwd_impl(dY, G, X, X32, gamma, mean, rstd, dX, dG, dgamma, dbeta, dbias, accumulate, M, d, p_drop, seed, seed_dev, site,
                         row_offset, workspace, nullptr, st);
}
extern "C" int hriemo_add_ln_bwd_rows(const void* dY, const void* G, const void* X, const float* X32, const float* gamma, const float* mean,
                                      const float* rstd, void* dX, void* dG, float* dgamma, float* dbeta, float* dbias, int accumulate,
                                      int M, int d, float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                                      unsigned site, long row_offset, float* workspace, const long long* row_index, hipStream_t st) {
  return add_ln_bwd_impl(dY, G, X, X32, gamma, mean, rstd, dX, dG, dgamma, dbeta, dbias, accumulate, M, d, p_drop, seed, seed_dev, site,
                         row_offset, workspace, row_index, st);
}

extern "C" int hriemo_add_ln_bwd_partial_rows(int M, int d) { (void)d; return row_grid(M, lnb_cap(d)); }

static int colsum_slices(int M, int N) {
  const int ncg = (N / 8 + 63) / 64;
  int slices = 2048 / ncg;
  if (slices > (M + 15) / 16) slices = (M + 15) / 16;
  if (slices < 1) slices = 1;
  return slices;
}
extern "C" long hriemo_colsum_workspace_bytes(int M, int N) { return ((long)colsum_slices(M, N) + 64) * N * 4; }

extern "C" int hriemo_colsum_bf16(const void* X, long ldx, int M, int N, float* out, int accumulate, float* workspace, hipStream_t st) {
  HRIEMO_CHECK(M > 0 && N > 0 && N % 8 == 0 && ldx % 8 == 0, "colsum: bad shape M=%d N=%d ld=%ld", M, N, ldx);
  HRIEMO_CHECK(workspace != nullptr, "colsum: workspace required");
  const int ncg = (N / 8 + 63) / 64;
  const int slices = colsum_slices(M, N);
  const int rps = (M + slices - 1) / slices;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(ncg, slices), dim3(64), 0, st, (const bf16_t*)X, ldx, M, N, rps, workspace);
  if (out == nullptr) { HRIEMO_LAUNCH_CHECK("colsum_partial_kernel"); return 0; }   // [slices][N] partials left for hriemo_colreduce_batch
  ReduceOut ro; ro.o[0] = out; ro.o[1] = ro.o[2] = nullptr;
  launch_colreduce(workspace, (long)N, slices, ro, N, 1, accumulate, workspace + (long)slices * N, st);
  HRIEMO_LAUNCH_CHECK("colsum");
  return 0;
}

extern "C" int hriemo_colsum_partial_rows(int M, int N) { return colsum_slices(M, N); }

// job records travel host -> device as kernel ARGUMENTS (48 records = 3 KB per launch): no pinned staging buffer,
// nothing a stream capture could object to, and the values are baked into a captured graph
struct JobChunk { long long v[48 * 8]; };
__global__ void upload_jobs_kernel(const JobChunk c, long long* __restrict__ dst, int n) {
  for (int i = threadIdx.x; i < n * 8; i += blockDim.x) dst[i] = c.v[i];
}

extern "C" int hriemo_colreduce_batch(const void* jobs_host, int njobs, void* jobs_dev, int nblocks, hipStream_t st) {
  HRIEMO_CHECK(jobs_host != nullptr && jobs_dev != nullptr && njobs > 0 && nblocks > 0, "colreduce_batch: empty job table");
  for (int j0 = 0; j0 < njobs; j0 += 48) {
    JobChunk c;
    const int n = njobs - j0 < 48 ? njobs - j0 : 48;
    memcpy(c.v, (const long long*)jobs_host + (long)j0 * 8, (size_t)n * 64);
    hipLaunchKernelGGL(upload_jobs_kernel, dim3(1), dim3(256), 0, st, c, (long long*)jobs_dev + (long)j0 * 8, n);
  }
  HRIEMO_LAUNCH_CHECK("upload_jobs_kernel");
  hriemo_prof_begin(HP_ROWOPS, st);
  hipLaunchKernelGGL(colreduce_batch_kernel, dim3(nblocks), dim3(256), 0, st, (const long long*)jobs_dev, njobs);
  HRIEMO_LAUNCH_CHECK("colreduce_batch_kernel");
  hriemo_prof_end(HP_ROWOPS, st, 0.0);
  return 0;
}

extern "C" int hriemo_cast_f32_to_bf16(const float* src, void* dst, long n, hipStream_t st) {
  HRIEMO_CHECK(n > 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "cast: empty or unaligned");
  long g = ((n >> 3) + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((int)g), dim3(256), 0, st, src, (bf16_t*)dst, n);
  HRIEMO_LAUNCH_CHECK("cast_f32_bf16_kernel");
  return 0;
}
// Many fp32 -> bf16 casts in ONE launch: the bf16 shadows of every weight matrix a step uses (46 matrices at cfg 2) were 46
// dependent ~5 us launches, 16 of them a serial chain in front of the step's first GEMM (profiles/r03_step_timeline.txt).
// Job j = (src, dst, n elements, first block): the table travels as a kernel argument (capture-safe, no staging buffer); a
// block finds its job by a linear scan (<= 64 entries) and casts a 2048-element span of it.
struct CastJobs { const float* src[64]; void* dst[64]; long n[64]; int first[65]; int njobs; unsigned long long f32_copy; };
__global__ __launch_bounds__(256) void cast_f32_bf16_batch_kernel(const CastJobs J) {
  int j = 0;
  while (j + 1 < J.njobs && (int)blockIdx.x >= J.first[j + 1]) ++j;
  const long n = J.n[j], base = ((long)blockIdx.x - J.first[j]) * 2048 + (long)threadIdx.x * 8;
  const float* src = J.src[j];
  if ((J.f32_copy >> j) & 1ull) {            // job kind 1: fp32 -> fp32 (bias slices into a concatenated vector)
    float* dst = (float*)J.dst[j];
    if (base + 8 <= n) {
      *(f32x4*)(dst + base) = *(const f32x4*)(src + base);
      *(f32x4*)(dst + base + 4) = *(const f32x4*)(src + base + 4);
    } else {
      for (long e = base; e < n; ++e) dst[e] = src[e];
    }
    return;
  }
  bf16_t* dst = (bf16_t*)J.dst[j];
  if (base + 8 <= n) {
    const f32x4 a = *(const f32x4*)(src + base), b = *(const f32x4*)(src + base + 4);
    float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    *(bf16x8*)(dst + base) = f32_to_bf8(f);
  } else {
    for (long e = base; e < n; ++e) dst[e] = (bf16_t)src[e];
  }
}
static int cast_batch_impl(const long long* h, int njobs, int rec, hipStream_t st) {
  for (int j0 = 0; j0 < njobs; j0 += 64) {
    CastJobs J;
    const int nj = njobs - j0 < 64 ? njobs - j0 : 64;
    int blocks = 0;
    J.f32_copy = 0ull;
    for (int j = 0; j < nj; ++j) {
      const long long* r = h + (long)(j0 + j) * rec;
      HRIEMO_CHECK(r[2] > 0 && (r[0] % 16) == 0 && (r[1] % 16) == 0, "cast_batch: job %d empty or unaligned", j0 + j);
      HRIEMO_CHECK(rec == 3 || r[3] == 0 || r[3] == 1, "cast_batch: job %d has kind %lld (0 = fp32 -> bf16, 1 = fp32 copy)", j0 + j, r[3]);
      J.src[j] = (const float*)r[0]; J.dst[j] = (void*)r[1]; J.n[j] = (long)r[2];
      if (rec == 4 && r[3] == 1) J.f32_copy |= 1ull << j;
      J.first[j] = blocks;
      blocks += (int)((r[2] + 2047) / 2048);
    }
    J.first[nj] = blocks; J.njobs = nj;
    hriemo_prof_begin(HP_ROWOPS, st);
    hipLaunchKernelGGL(cast_f32_bf16_batch_kernel, dim3(blocks), dim3(256), 0, st, J);
    HRIEMO_LAUNCH_CHECK("cast_f32_bf16_batch_kernel");
    hriemo_prof_end(HP_ROWOPS, st, 0.0);
  }
  return 0;
}
// jobs_host: njobs records of 3 x int64 {src, dst, n}; src / dst 16-byte aligned
extern "C" int hriemo_cast_f32_to_bf16_batch(const void* jobs_host, int njobs, hipStream_t st) {
  HRIEMO_CHECK(jobs_host != nullptr && njobs > 0, "cast_batch: empty job table");
  return cast_batch_impl((const long long*)jobs_host, njobs, 3, st);
}
// jobs_host: njobs records of 4 x int64 {src, dst, n, kind}; kind 0 = fp32 -> bf16, 1 = fp32 -> fp32.  One launch refreshes a
// shared projection's concatenated weight shadow AND its concatenated bias vector (before: two casts, a cat and a copy).
extern "C" int hriemo_cast_copy_batch(const void* jobs_host, int njobs, hipStream_t st) {
  HRIEMO_CHECK(jobs_host != nullptr && njobs > 0, "cast_copy_batch: empty job table");
  return cast_batch_impl((const long long*)jobs_host, njobs, 4, st);
}
extern "C" int hriemo_cast_bf16_to_f32(const void* src, float* dst, long n, hipStream_t st) {
  HRIEMO_CHECK(n > 0, "cast: empty");
  long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((int)g), dim3(256), 0, st, (const bf16_t*)src, dst, n);
  HRIEMO_LAUNCH_CHECK("cast_bf16_f32_kernel");
  return 0;
}

extern "C" int hriemo_dropout_bf16(const void* X, void* Y, long M, int N, float p_drop, unsigned long long seed,
                                   const unsigned long long* seed_dev, unsigned site, long row_offset, hipStream_t st) {
  HRIEMO_CHECK(M > 0 && N > 0 && N % 8 == 0, "dropout: bad shape");
  RowDrop dr = row_drop(p_drop, seed, seed_dev, site);
  long g = (M * (N / 8) + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(dropout_kernel, dim3((int)g), dim3(256), 0, st, (const bf16_t*)X, (bf16_t*)Y, M, N, dr, row_offset);
  HRIEMO_LAUNCH_CHECK("dropout_kernel");
  return 0;
}

// ---- packed (varlen) sequences: gather / scatter between the padded [B, L, d] layout and the packed [n_rows, d] one -----------
// Sequence b owns packed rows cu[b] .. cu[b+1]-1 (its first cu[b+1]-cu[b] positions of the padded layout; the collate pads at the
// end, train_fusion_seq_level_decoder.py:191-232).  The lengths are DEVICE data: a captured graph serves every batch whose packed
// rows fit n_rows.  Packed rows cu[B] .. n_rows-1 belong to no sequence (the host rounds the row count up to a bucket and treats
// them as one extra sequence): they are written as zeros, so every later kernel sees finite values there and their gradients are
// exactly zero.  One launch moves the bf16 tensor and its fp32 twin (either may be NULL) and emits the padded row of every packed
// row (row_index of hriemo_add_ln_*_rows: the dropout hash stays keyed by the padded position).  One wave per row.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4r;
__device__ __forceinline__ void copy_row16(const char* src, char* dst, int bytes, int lane) {
  for (int o = lane * 16; o < bytes; o += 64 * 16) *(u32x4r*)(dst + o) = src ? *(const u32x4r*)(src + o) : (u32x4r){0u, 0u, 0u, 0u};
}
__global__ __launch_bounds__(256) void pack_rows_kernel(const bf16_t* __restrict__ X16, const float* __restrict__ X32, const int* __restrict__ cu,
                                                        int B, int L, int d, int n_rows, bf16_t* __restrict__ P16, float* __restrict__ P32,
                                                        long long* __restrict__ rows) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  int lo = 0, hi = B;                       // largest b in [0, B] with cu[b] <= r  (b == B: beyond the last sequence)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (cu[mid] <= r) lo = mid; else hi = mid - 1;
  }
  const int b = lo, l = r - cu[b];
  const bool real = b < B && l < L;
  const long src = (long)b * L + l;
  if (rows != nullptr && lane == 0) rows[r] = src;
  if (P16 != nullptr) copy_row16(real ? (const char*)(X16 + src * d) : nullptr, (char*)(P16 + (long)r * d), d * 2, lane);
  if (P32 != nullptr) copy_row16(real ? (const char*)(X32 + src * d) : nullptr, (char*)(P32 + (long)r * d), d * 4, lane);
}
__global__ __launch_bounds__(256) void unpack_rows_kernel(const bf16_t* __restrict__ P16, const float* __restrict__ P32, const int* __restrict__ cu,
                                                          int B, int L, int d, bf16_t* __restrict__ Y16, float* __restrict__ Y32) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * L) return;
  const int b = (int)(row / L), l = (int)(row - (long)b * L);
  const int c0 = cu[b], len = cu[b + 1] - c0;
  const bool real = l < len;
  const long src = (long)c0 + l;
  if (Y16 != nullptr) copy_row16(real ? (const char*)(P16 + src * d) : nullptr, (char*)(Y16 + row * d), d * 2, lane);
  if (Y32 != nullptr) copy_row16(real ? (const char*)(P32 + src * d) : nullptr, (char*)(Y32 + row * d), d * 4, lane);
}
extern "C" int hriemo_pack_rows(const void* X16, const float* X32, const int* cu_seqlens, int B, int L, int d, int n_rows, void* P16,
                                float* P32, long long* row_index, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && L > 0 && d > 0 && d % 8 == 0 && n_rows > 0 && cu_seqlens != nullptr, "pack_rows: bad shape (B=%d L=%d d=%d n_rows=%d)", B, L, d, n_rows);
  HRIEMO_CHECK((X16 == nullptr) == (P16 == nullptr) && (X32 == nullptr) == (P32 == nullptr), "pack_rows: source and destination of a tensor must both be given or both be NULL");
  HRIEMO_CHECK(((uintptr_t)X16 % 16) == 0 && ((uintptr_t)X32 % 16) == 0 && ((uintptr_t)P16 % 16) == 0 && ((uintptr_t)P32 % 16) == 0, "pack_rows: unaligned operand");
  hriemo_prof_begin(HP_ROWOPS, st);
  hipLaunchKernelGGL(pack_rows_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, st, (const bf16_t*)X16, X32, cu_seqlens, B, L, d, n_rows, (bf16_t*)P16, P32, row_index);
  HRIEMO_LAUNCH_CHECK("pack_rows_kernel");
  hriemo_prof_end(HP_ROWOPS, st, (double)n_rows * d * ((X16 ? 4.0 : 0.0) + (X32 ? 8.0 : 0.0)));
  return 0;
}
extern "C" int hriemo_unpack_rows(const void* P16, const float* P32, const int* cu_seqlens, int B, int L, int d, void* Y16, float* Y32,
                                  hipStream_t st) {
  HRIEMO_CHECK(B > 0 && L > 0 && d > 0 && d % 8 == 0 && cu_seqlens != nullptr, "unpack_rows: bad shape (B=%d L=%d d=%d)", B, L, d);
  HRIEMO_CHECK((P16 == nullptr) == (Y16 == nullptr) && (P32 == nullptr) == (Y32 == nullptr), "unpack_rows: source and destination of a tensor must both be given or both be NULL");
  HRIEMO_CHECK(((uintptr_t)P16 % 16) == 0 && ((uintptr_t)P32 % 16) == 0 && ((uintptr_t)Y16 % 16) == 0 && ((uintptr_t)Y32 % 16) == 0, "unpack_rows: unaligned operand");
  const long rows = (long)B * L;
  hriemo_prof_begin(HP_ROWOPS, st);
  hipLaunchKernelGGL(unpack_rows_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, st, (const bf16_t*)P16, P32, cu_seqlens, B, L, d, (bf16_t*)Y16, Y32);
  HRIEMO_LAUNCH_CHECK("unpack_rows_kernel");
  hriemo_prof_end(HP_ROWOPS, st, (double)rows * d * ((P16 ? 4.0 : 0.0) + (P32 ? 8.0 : 0.0)));
  return 0;
}

extern "C" int hriemo_expand_rows(const float* q, void* out, float* out32, int B, long n, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && n > 0, "expand: empty");
  long g = ((long)B * n + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(expand_rows_kernel, dim3((int)g), dim3(256), 0, st, q, (bf16_t*)out, out32, B, n);
  HRIEMO_LAUNCH_CHECK("expand_rows_kernel");
  return 0;
}
extern "C" int hriemo_seed_bump(unsigned long long* seed_dev, hipStream_t st) {
  HRIEMO_CHECK(seed_dev != nullptr, "seed_bump: no seed word");
  hipLaunchKernelGGL(seed_bump_kernel, dim3(1), dim3(64), 0, st, seed_dev);
  HRIEMO_LAUNCH_CHECK("seed_bump_kernel");
  return 0;
}

extern "C" int hriemo_rowdot_fwd(const void* Z, const float* Z32, const float* w, const float* b, float* out, int M, int d, hipStream_t st) {
  if (check_rows(M, d)) return 1;
  hipLaunchKernelGGL(rowdot_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, st, (const bf16_t*)Z, Z32, w, b, out, M, d);
  HRIEMO_LAUNCH_CHECK("rowdot_fwd_kernel");
  return 0;
}
extern "C" int hriemo_rowdot_bwd(const float* dl, const void* Z, const float* Z32, const float* w, void* dZ, float* dw, float* db,
                                 int accumulate, int M, int d, hipStream_t st) {
  if (check_rows(M, d)) return 1;
  hipLaunchKernelGGL(rowdot_bwd_kernel, dim3((d + 31) / 32), dim3(ROWDOT_GROUPS * 32), 0, st, dl, (const bf16_t*)Z, Z32, w, (bf16_t*)dZ, dw, db, M, d, accumulate);
  HRIEMO_LAUNCH_CHECK("rowdot_bwd_kernel");
  return 0;
}

// ------------------------------------------------------------------ legacy scalar beta gate (models/beta_gate.py)
// pooled[b, :] = sum over valid rows of X[b, :, :] / max(#valid, 1)   (beta_gate.py:6-32); cnt[b] = max(#valid, 1)
__global__ __launch_bounds__(256) void masked_mean_fwd_kernel(const bf16_t* __restrict__ X, const uint8_t* __restrict__ mask,
                                                              float* __restrict__ pooled, float* __restrict__ cnt, int L, int d) {
  __shared__ float red[4][64][8];
  __shared__ float cred[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, ch = blockIdx.x * 64 + lane, nchunk = d >> 3;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float c = 0.f;
  for (int l = wave; l < L; l += 4) {
    const bool valid = mask == nullptr || mask[(long)b * L + l] == 0;
    if (lane == 0 && valid) c += 1.f;
    if (valid && ch < nchunk) {
      const bf16x8 v = *(const bf16x8*)(X + ((long)b * L + l) * d + ch * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[wave][lane][j] = acc[j];
  if (lane == 0) cred[wave] = c;
  __syncthreads();
  if (wave == 0) {
    const float n = fmaxf(cred[0] + cred[1] + cred[2] + cred[3], 1.f);
    if (ch < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        pooled[(long)b * d + ch * 8 + j] = (red[0][lane][j] + red[1][lane][j] + red[2][lane][j] + red[3][lane][j]) / n;
    }
    if (lane == 0 && blockIdx.x == 0) cnt[b] = n;
  }
}

// out[b] = sum of the n floats of row b (dbeta of the scalar gate from the fuse_bwd_dw partials)
__global__ __launch_bounds__(256) void rowsum_f32_kernel(const float* __restrict__ x, float* __restrict__ out, long n) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) s += x[(long)b * n + i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[b] = red[0] + red[1] + red[2] + red[3];
}

// dX[b, l, :] = coef(b) * dH[b, l, :] (rows l < Lf) + valid(b, l) * dpool[b, :] / cnt[b],  coef = beta or 1 - beta
// (backward of h_fusion = beta*h_a[:, :L] + (1-beta)*h_t[:, :L] and of masked_mean; beta_gate.py:97-112)
__global__ void scalar_gate_dx_kernel(const bf16_t* __restrict__ dH, int Lf, const float* __restrict__ beta, int is_a,
                                      const float* __restrict__ dpool, const float* __restrict__ cnt, const uint8_t* __restrict__ mask,
                                      bf16_t* __restrict__ dX, int B, int L, int d) {
  const int nch = d >> 3;
  const long nv = (long)B * L * nch;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(v % nch);
    const long row = v / nch;
    const int b = (int)(row / L), l = (int)(row - (long)b * L);
    const float coef = is_a ? beta[b] : 1.f - beta[b];
    const bool valid = mask == nullptr || mask[row] == 0;
    const float ic = valid ? 1.f / cnt[b] : 0.f;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = ic * dpool[(long)b * d + ch * 8 + j];
    if (l < Lf) {
      const bf16x8 g = *(const bf16x8*)(dH + ((long)b * Lf + l) * d + ch * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += coef * (float)g[j];
    }
    *(bf16x8*)(dX + row * d + ch * 8) = f32_to_bf8(o);
  }
}

// gate input of the legacy scalar gate from already pooled means (models/beta_gate.py:82-93): [a, t, |a-t|, a*t] -> bf16 [B, 4d]
__global__ __launch_bounds__(256) void gate_input_pooled_kernel(const float* __restrict__ a_pool, const float* __restrict__ t_pool,
                                                                bf16_t* __restrict__ gin, int d) {
  const int b = blockIdx.x;
  bf16_t* g = gin + (long)b * 4 * d;
  for (int c = threadIdx.x; c < d; c += 256) {
    const float a = a_pool[(long)b * d + c], t = t_pool[(long)b * d + c];
    g[c] = (bf16_t)a; g[d + c] = (bf16_t)t; g[2 * d + c] = (bf16_t)fabsf(a - t); g[3 * d + c] = (bf16_t)(a * t);
  }
}

// ------------------------------------------------------------------ optimizer (trainer step, off the timed path)
// partial[blockIdx.x] = sum of squares of this block's grid-stride share (fixed grid: deterministic)
__global__ __launch_bounds__(256) void sumsq_f32_kernel(const float* __restrict__ x, long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  const long nv = n >> 2;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (long)gridDim.x * 256) {
    const f32x4 t = *(const f32x4*)(x + v * 4);
    s += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// clip_grad_norm_(max_norm) + AdamW over flat, identically laid out fp32 buffers (parameters, gradients, moments):
// torch.nn.utils.clip_grad_norm_ (coef = min(1, max_norm / (norm + 1e-6))) followed by torch.optim.AdamW's update
// (train_fusion_seq_level_decoder.py:332-334).  norm2 is read from device memory: no host round trip.
__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                         float wd, float bc1, float bc2, float max_norm, const float* __restrict__ norm2) {
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(1.f, max_norm / (sqrtf(norm2[0]) + 1e-6f));
  const float step = lr / bc1, isb2 = 1.f / sqrtf(bc2);
  const long nv = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    f32x4 pp = *(const f32x4*)(p + i * 4), mm = *(const f32x4*)(m + i * 4), vv = *(const f32x4*)(v + i * 4);
    const f32x4 gg = *(const f32x4*)(g + i * 4) * coef;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pp[e] *= 1.f - lr * wd;
      mm[e] += (gg[e] - mm[e]) * (1.f - b1);
      vv[e] = vv[e] * b2 + gg[e] * gg[e] * (1.f - b2);
      pp[e] -= step * mm[e] / (sqrtf(vv[e]) * isb2 + eps);
    }
    *(f32x4*)(p + i * 4) = pp; *(f32x4*)(m + i * 4) = mm; *(f32x4*)(v + i * 4) = vv;
  }
}

// ---- beta gate ----
extern "C" int hriemo_pool_chunks(int L) { return (L + 31) / 32; }

extern "C" int hriemo_ln_pool_fwd(const void* X, const float* X32, const unsigned char* mask, const float* gamma, const float* beta, void* Yn,
                                  float* mean, float* rstd, float* partials, int B, int L, int Lkeep, int d, float eps,
                                  hipStream_t st) {
  if (check_rows(B * L, d)) return 1;
  HRIEMO_CHECK(Lkeep >= 0 && Lkeep <= L, "ln_pool_fwd: Lkeep=%d out of range (L=%d)", Lkeep, L);
  const int nc = (L + 31) / 32;
  hriemo_prof_begin(HP_ROWOPS, st);
  // (next-row prefetch for d <= 1024, where it fits the registers)
#define CALL(N)                                                                                                                  \
  if ((N) <= 2)                                                                                                                  \
    hipLaunchKernelGGL((ln_pool_fwd_kernel<N, ((N) <= 2)>), dim3(nc, B), dim3(256), d * 4, st, (const bf16_t*)X, X32, mask, gamma, beta, (bf16_t*)Yn, mean, rstd, partials, L, Lkeep, d, eps); \
  else                                                                                                                           \
    hipLaunchKernelGGL((ln_pool_fwd_kernel<N, false>), dim3(nc, B), dim3(256), d * 4, st, (const bf16_t*)X, X32, mask, gamma, beta, (bf16_t*)Yn, mean, rstd, partials, L, Lkeep, d, eps)
  DISPATCH_NCH(d, CALL)
#undef CALL
  HRIEMO_LAUNCH_CHECK("ln_pool_fwd_kernel");
  hriemo_prof_end(HP_ROWOPS, st, ((double)B * L + (double)B * Lkeep) * d * 2);
  return 0;
}

// the gate's two LayerNorm + pool steps from one launch (side 0 = audio, side 1 = text; same d, B, eps): see ln_pool_fwd_pair_kernel
extern "C" int hriemo_ln_pool_pair_supported(int d) { return d % 8 == 0 && d <= 1024 ? 1 : 0; }
extern "C" int hriemo_ln_pool_fwd_pair(const void* Xa, const float* Xa32, const unsigned char* mask_a, const float* gamma_a, const float* beta_a,
                                       void* Yna, float* mean_a, float* rstd_a, float* partials_a, int La,
                                       const void* Xt, const float* Xt32, const unsigned char* mask_t, const float* gamma_t, const float* beta_t,
                                       void* Ynt, float* mean_t, float* rstd_t, float* partials_t, int Lt,
                                       int B, int Lkeep, int d, float eps, hipStream_t st) {
  if (check_rows(B * La, d) || check_rows(B * Lt, d)) return 1;
  HRIEMO_CHECK(Lkeep >= 0 && Lkeep <= La && Lkeep <= Lt, "ln_pool_fwd_pair: Lkeep=%d out of range (La=%d, Lt=%d)", Lkeep, La, Lt);
  HRIEMO_CHECK(hriemo_ln_pool_pair_supported(d), "ln_pool_fwd_pair: d=%d (pairs are built for d <= 1024; use two hriemo_ln_pool_fwd calls)", d);
  PoolFwdPair p;
  p.s[0] = PoolFwdSide{(const bf16_t*)Xa, Xa32, mask_a, gamma_a, beta_a, (bf16_t*)Yna, mean_a, rstd_a, partials_a, La, Lkeep, (La + 31) / 32};
  p.s[1] = PoolFwdSide{(const bf16_t*)Xt, Xt32, mask_t, gamma_t, beta_t, (bf16_t*)Ynt, mean_t, rstd_t, partials_t, Lt, Lkeep, (Lt + 31) / 32};
  const int nc = p.s[0].nc + p.s[1].nc;
  hriemo_prof_begin(HP_ROWOPS, st);
  if (d <= 512) hipLaunchKernelGGL((ln_pool_fwd_pair_kernel<1, true>), dim3(nc, B), dim3(256), d * 4, st, p, d, eps);
  else hipLaunchKernelGGL((ln_pool_fwd_pair_kernel<2, true>), dim3(nc, B), dim3(256), d * 4, st, p, d, eps);
  HRIEMO_LAUNCH_CHECK("ln_pool_fwd_pair_kernel");
  hriemo_prof_end(HP_ROWOPS, st, ((double)B * (La + Lt) + 2.0 * B * Lkeep) * d * 2);
  return 0;
}

extern "C" int hriemo_gate_input(const float* partials_a, const float* partials_t, const unsigned char* mask_a,
                                 const unsigned char* mask_t, int B, int La, int Lt, int d, void* gate_in, float* a_pool,
                                 float* t_pool, float* cnt, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && d > 0, "gate_input: empty");
  hipLaunchKernelGGL(gate_input_kernel, dim3(B), dim3(256), 0, st, partials_a, (La + 31) / 32, partials_t, (Lt + 31) / 32, mask_a,
                     mask_t, La, Lt, d, (bf16_t*)gate_in, a_pool, t_pool, cnt);
  HRIEMO_LAUNCH_CHECK("gate_input_kernel");
  return 0;
}

extern "C" int hriemo_sigmoid_beta(const float* pre, float* w, float* beta, int B, int d, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && d > 0, "sigmoid_beta: empty");
  hipLaunchKernelGGL(sigmoid_beta_kernel, dim3(B), dim3(256), 0, st, pre, w, beta, d);
  HRIEMO_LAUNCH_CHECK("sigmoid_beta_kernel");
  return 0;
}

extern "C" int hriemo_masked_mean_fwd(const void* X, const unsigned char* mask, float* pooled, float* cnt, int B, int L, int d,
                                      hipStream_t st) {
  if (check_rows(B * L, d)) return 1;
  hriemo_prof_begin(HP_ROWOPS, st);
  hipLaunchKernelGGL(masked_mean_fwd_kernel, dim3((d / 8 + 63) / 64, B), dim3(256), 0, st, (const bf16_t*)X, mask, pooled, cnt, L, d);
  HRIEMO_LAUNCH_CHECK("masked_mean_fwd_kernel");
  hriemo_prof_end(HP_ROWOPS, st, 1.0 * B * L * d * 2);
  return 0;
}

extern "C" int hriemo_gate_input_pooled(const float* a_pool, const float* t_pool, void* gate_in, int B, int d, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && d > 0, "gate_input_pooled: empty");
  hipLaunchKernelGGL(gate_input_pooled_kernel, dim3(B), dim3(256), 0, st, a_pool, t_pool, (bf16_t*)gate_in, d);
  HRIEMO_LAUNCH_CHECK("gate_input_pooled_kernel");
  return 0;
}

extern "C" int hriemo_rowsum_f32(const float* x, float* out, int B, long n, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && n > 0, "rowsum_f32: empty problem");
  hipLaunchKernelGGL(rowsum_f32_kernel, dim3(B), dim3(256), 0, st, x, out, n);
  HRIEMO_LAUNCH_CHECK("rowsum_f32_kernel");
  return 0;
}

extern "C" int hriemo_scalar_gate_dx(const void* dH, int Lf, const float* beta, int is_a, const float* dpool, const float* cnt,
                                     const unsigned char* mask, void* dX, int B, int L, int d, hipStream_t st) {
  if (check_rows(B * L, d)) return 1;
  HRIEMO_CHECK(Lf >= 0 && Lf <= L, "scalar_gate_dx: Lf=%d out of range (L=%d)", Lf, L);
  long g = ((long)B * L * (d / 8) + 255) / 256;
  if (g > 8192) g = 8192;
  hriemo_prof_begin(HP_ROWOPS, st);
  hipLaunchKernelGGL(scalar_gate_dx_kernel, dim3((int)g), dim3(256), 0, st, (const bf16_t*)dH, Lf, beta, is_a, dpool, cnt, mask,
                     (bf16_t*)dX, B, L, d);
  HRIEMO_LAUNCH_CHECK("scalar_gate_dx_kernel");
  hriemo_prof_end(HP_ROWOPS, st, 2.0 * B * L * d * 2);
  return 0;
}

extern "C" int hriemo_sumsq_f32(const float* x, long n, float* partial, int nblocks, hipStream_t st) {
  HRIEMO_CHECK(n > 0 && n % 4 == 0 && nblocks > 0 && nblocks <= 4096, "sumsq_f32: n=%ld must be a positive multiple of 4, 0 < nblocks <= 4096", n);
  HRIEMO_CHECK(((uintptr_t)x % 16) == 0, "sumsq_f32: unaligned buffer");
  hipLaunchKernelGGL(sumsq_f32_kernel, dim3(nblocks), dim3(256), 0, st, x, n, partial);
  HRIEMO_LAUNCH_CHECK("sumsq_f32_kernel");
  return 0;
}

extern "C" int hriemo_adamw_flat(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, int step, float max_norm, const float* norm2, hipStream_t st) {
  HRIEMO_CHECK(n > 0 && n % 4 == 0 && step >= 1, "adamw_flat: n=%ld must be a positive multiple of 4 and step >= 1", n);
  HRIEMO_CHECK(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0, "adamw_flat: unaligned buffer");
  HRIEMO_CHECK(max_norm <= 0.f || norm2 != nullptr, "adamw_flat: clipping needs the squared gradient norm");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  long gsz = (n / 4 + 255) / 256;
  if (gsz > 4096) gsz = 4096;
  hipLaunchKernelGGL(adamw_flat_kernel, dim3((int)gsz), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                     max_norm, norm2);
  HRIEMO_LAUNCH_CHECK("adamw_flat_kernel");
  return 0;
}

extern "C" int hriemo_fuse_fwd(const float* w, const void* A, const void* T, void* H, int B, int L, int d, hipStream_t st) {
  if (check_rows(B * L, d)) return 1;
  long g = ((long)B * L * (d / 8) + 255) / 256;
  if (g > 8192) g = 8192;
  hriemo_prof_begin(HP_ROWOPS, st);
  hipLaunchKernelGGL(fuse_fwd_kernel, dim3((int)g), dim3(256), 0, st, w, (const bf16_t*)A, (const bf16_t*)T, (bf16_t*)H, B, L, d);
  HRIEMO_LAUNCH_CHECK("fuse_fwd_kernel");
  hriemo_prof_end(HP_ROWOPS, st, 3.0 * B * L * d * 2);
  return 0;
}

extern "C" int hriemo_fuse_bwd_dw(const void* dH, const void* A, const void* T, float* partials, int B, int L, int d,
                                  hipStream_t st) {
  if (check_rows(B * L, d)) return 1;
  const int nc = (L + 31) / 32;
#define CALL(N) hipLaunchKernelGGL((fuse_bwd_dw_kernel<N>), dim3(nc, B), dim3(256), d * 4, st, (const bf16_t*)dH, (const bf16_t*)A, (const bf16_t*)T, partials, L, d)
  DISPATCH_NCH(d, CALL)
#undef CALL
  HRIEMO_LAUNCH_CHECK("fuse_bwd_dw_kernel");
  return 0;
}

extern "C" int hriemo_gate_dpre(const float* partials, int L, const float* dbeta, const float* w, void* dpre, int B, int d,
                                hipStream_t st) {
  HRIEMO_CHECK(B > 0 && d > 0, "gate_dpre: empty");
  hipLaunchKernelGGL(gate_dpre_kernel, dim3(B), dim3(256), 0, st, partials, (L + 31) / 32, dbeta, w, (bf16_t*)dpre, d);
  HRIEMO_LAUNCH_CHECK("gate_dpre_kernel");
  return 0;
}

extern "C" int hriemo_gate_input_bwd(const void* dgin, const float* a_pool, const float* t_pool, const float* cnt, float* da,
                                     float* dt, int B, int d, hipStream_t st) {
  HRIEMO_CHECK(B > 0 && d > 0, "gate_input_bwd: empty");
  hipLaunchKernelGGL(gate_input_bwd_kernel, dim3(B), dim3(256), 0, st, (const bf16_t*)dgin, a_pool, t_pool, cnt, da, dt, d);
  HRIEMO_LAUNCH_CHECK("gate_input_bwd_kernel");
  return 0;
}

extern "C" int hriemo_ln_pool_bwd_chunks(int L) { return (L + POOL_BWD_ROWS - 1) / POOL_BWD_ROWS; }
extern "C" long hriemo_ln_pool_bwd_workspace_bytes(int B, int L, int d) { return ((long)B * hriemo_ln_pool_bwd_chunks(L) * 2 * d + 64L * 2 * d) * 4; }

extern "C" int hriemo_ln_pool_bwd(const void* dH, int Lf, const float* w, int is_a, const float* dpool, const unsigned char* mask,
                                  const void* X, const float* X32, const float* gamma, const float* mean, const float* rstd, void* dX,
                                  float* dgamma, float* dbeta, int accumulate, int B, int L, int d, float* workspace, hipStream_t st) {
  if (check_rows(B * L, d)) return 1;
  HRIEMO_CHECK(workspace != nullptr && Lf <= L && (dgamma == nullptr) == (dbeta == nullptr), "ln_pool_bwd: bad arguments");
  const int nc = hriemo_ln_pool_bwd_chunks(L);
  hriemo_prof_begin(HP_ROWOPS, st);
#define CALL(N) hipLaunchKernelGGL((ln_pool_bwd_kernel<N>), dim3(nc, B), dim3(256), ((N) <= 2 ? 5 : 2) * d * 4, st, (const bf16_t*)dH, Lf, w, is_a, dpool, mask, (const bf16_t*)X, X32, gamma, mean, rstd, (bf16_t*)dX, workspace, L, d)
  DISPATCH_NCH(d, CALL)
#undef CALL
  HRIEMO_LAUNCH_CHECK("ln_pool_bwd_kernel");
  hriemo_prof_end(HP_ROWOPS, st, (2.0 * B * L + (double)B * Lf) * d * 2);
  if (dgamma == nullptr) return 0;          // partial sums stay in `workspace` ([B * hriemo_ln_pool_bwd_chunks(L)][2d]) for the launch-boundary reduce
  float* scratch = workspace + (long)B * nc * 2 * d;
  ReduceOut ro; ro.o[0] = dgamma; ro.o[1] = dbeta; ro.o[2] = nullptr;
  launch_colreduce(workspace, (long)2 * d, B * nc, ro, d, 2, accumulate, scratch, st);
  HRIEMO_LAUNCH_CHECK("colreduce_kernel");
  return 0;
}

// both modalities' gate backward from one launch (side 0 = audio: coefficient w, side 1 = text: 1 - w).  dgamma / dbeta as in the
// single call: finished here when given, else the partial sums stay in each side's workspace for the launch-boundary reduce
extern "C" int hriemo_ln_pool_bwd_pair(const void* dH, int Lf, const float* w,
                                       const float* dpool_a, const unsigned char* mask_a, const void* Xa, const float* Xa32, const float* gamma_a,
                                       const float* mean_a, const float* rstd_a, void* dXa, float* dgamma_a, float* dbeta_a, int La, float* workspace_a,
                                       const float* dpool_t, const unsigned char* mask_t, const void* Xt, const float* Xt32, const float* gamma_t,
                                       const float* mean_t, const float* rstd_t, void* dXt, float* dgamma_t, float* dbeta_t, int Lt, float* workspace_t,
                                       int accumulate, int B, int d, hipStream_t st) {
  if (check_rows(B * La, d) || check_rows(B * Lt, d)) return 1;
  HRIEMO_CHECK(workspace_a != nullptr && workspace_t != nullptr && workspace_a != workspace_t && Lf <= La && Lf <= Lt, "ln_pool_bwd_pair: bad arguments");
  HRIEMO_CHECK(hriemo_ln_pool_pair_supported(d), "ln_pool_bwd_pair: d=%d (pairs are built for d <= 1024; use two hriemo_ln_pool_bwd calls)", d);
  HRIEMO_CHECK((dgamma_a == nullptr) == (dbeta_a == nullptr) && (dgamma_t == nullptr) == (dbeta_t == nullptr) && (dgamma_a == nullptr) == (dgamma_t == nullptr),
               "ln_pool_bwd_pair: the four LayerNorm gradients come together or not at all");
  PoolBwdPair p;
  p.s[0] = PoolBwdSide{dpool_a, mask_a, (const bf16_t*)Xa, Xa32, gamma_a, mean_a, rstd_a, (bf16_t*)dXa, workspace_a, La, hriemo_ln_pool_bwd_chunks(La)};
  p.s[1] = PoolBwdSide{dpool_t, mask_t, (const bf16_t*)Xt, Xt32, gamma_t, mean_t, rstd_t, (bf16_t*)dXt, workspace_t, Lt, hriemo_ln_pool_bwd_chunks(Lt)};
  const int nc = p.s[0].nc + p.s[1].nc;
  hriemo_prof_begin(HP_ROWOPS, st);
  if (d <= 512) hipLaunchKernelGGL((ln_pool_bwd_pair_kernel<1>), dim3(nc, B), dim3(256), 5 * d * 4, st, (const bf16_t*)dH, Lf, w, p, d);
  else hipLaunchKernelGGL((ln_pool_bwd_pair_kernel<2>), dim3(nc, B), dim3(256), 5 * d * 4, st, (const bf16_t*)dH, Lf, w, p, d);
  HRIEMO_LAUNCH_CHECK("ln_pool_bwd_pair_kernel");
  hriemo_prof_end(HP_ROWOPS, st, (2.0 * B * (La + Lt) + 2.0 * B * Lf) * d * 2);
  if (dgamma_a == nullptr) return 0;
  for (int side = 0; side < 2; ++side) {
    float* ws = side ? workspace_t : workspace_a;
    float* scratch = ws + (long)B * p.s[side].nc * 2 * d;
    ReduceOut ro; ro.o[0] = side ? dgamma_t : dgamma_a; ro.o[1] = side ? dbeta_t : dbeta_a; ro.o[2] = nullptr;
    launch_colreduce(ws, (long)2 * d, B * p.s[side].nc, ro, d, 2, accumulate, scratch, st);
    HRIEMO_LAUNCH_CHECK("colreduce_kernel");
  }
  return 0;
}
