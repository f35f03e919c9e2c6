// Flash-style multi-head attention core for gfx950 (the part nn.MultiheadAttention hides between
// the in- and out-projection; reference call sites models/cross_modal_block_tacfn.py:74-117 and
// models/emotion_decoder.py:42-54).  Q/K/V are read in place from the projection outputs
// ([rows, ld] with the head at column h*HD), O is written head-concatenated for the out-projection.
//
// Layout choices (MFMA 16x16x32 bf16, wave64):
//  * forward / dQ: scores are produced TRANSPOSED, S^T = K.Q^T, so one lane owns one query row
//    (col = lane&15) and 4 keys per 16-key subtile; the row max/sum needs two xor-shuffles only,
//    and P (resp. dS) feeds the next MFMA straight from registers as the B operand with the
//    permuted k-order key = 16*(j>>2) + 4*(lane>>4) + (j&3); the other operand (V^T resp. K^T)
//    is fetched in that same order with ds_read_b64_tr_b16 from the row-major LDS tile.
//  * dK/dV: scores are produced un-transposed (key on the lane), so P / dS are again B operands
//    for the reductions over the query index; dK and dV of a key block live in registers while the
//    block sweeps all queries: no atomics, deterministic.
//  * LDS tiles are [rows][HD] bf16 with row stride HDP*2+32 bytes: conflict-free for both the
//    ds_read_b128 row reads and the transposed reads (8 consecutive rows -> 8 distinct 32-B windows).
//  * key-padding mask = additive -inf; a row whose keys are all PAD yields NaN like the reference.
//  * attention dropout is replayed from a counter hash (common.h), nothing is stored.
#include <stdlib.h>

#include "common.h"

struct AttnArgs {
  const bf16_t *Q, *K, *V;
  long ldq, ldk, ldv;
  bf16_t* O; long ldo;
  const bf16_t* dO; long lddo;
  bf16_t *dQ, *dK, *dV;
  long lddq, lddk, lddv;
  const uint8_t* kpm;
  float* lse;
  float* delta;
  float* probs;        // [B, Lq, Lk] head-averaged probabilities (export kernel only)
  float* csq;          // optional per-block column sums of the dQ tiles  [B*tiles_q, H*HD]      (in-proj bias gradient)
  float* cskv;         // optional per-block column sums of the dK | dV tiles [B*tiles_k, 2*H*HD]
  int B, H, Lq, Lk;
  float scale;
  uint32_t thr16; float inv_keep; uint64_t seed; uint32_t site; int b_offset;
  float l2ik, keepfrac;          // log2(1/(1-p)) and (1-p) (0 and 1 without dropout): scalars for the backward kernels
  const unsigned long long* seed_dev;
  // Dropout keep-mask as BITS, written once by the forward and read by both backward kernels instead of re-hashing
  // (3.3 MB per cross-attention site at cfg 2 against ~7 VALU operations per element and kernel).  One 64-bit word per
  // (batch, head, query, 64-key tile): bit 16*g + 4*n + r  <->  key 64*tile + 16*n + 4*g + r (the forward's register order:
  // lane group g holds keys 4g..4g+3 of each 16-key subtile n).  NULL: the backward replays the hash.
  unsigned long long* mbits;
  // Packed (varlen) sequences: rows of sample b are cu_q[b] .. cu_q[b+1]-1 of Q / O / dO / dQ and cu_k[b] .. cu_k[b+1]-1 of
  // K / V / dK / dV (int32 [B+1] each, both set or both NULL); Lq / Lk are then the LONGEST sequences (grid size, and the row
  // stride of the lse / delta / mask-word side buffers, which keep their padded [B,H,Lq] indexing).  Every length >= 1.
  const int* cu_q;
  const int* cu_k;
  // optional MX-fp8 copy of O for the out-projection GEMM (cfg 5's fp8 mode): bytes [B*Lq][ldoq] + E8M0 scales [H*HD/32][ldso]
  uint8_t* Oq; long ldoq;
  uint8_t* So; long ldso;
};

// Packed sequences: the kernels below index sample b's rows as b * L + r.  localize() turns the launch arguments into the view of
// ONE (batch, head): L becomes this sample's length and every row pointer is shifted so that b * L + r lands on the packed row
// cu[b] + r; the side buffers are shifted back onto their padded slots.  Blocks whose tile lies beyond the sample's length find
// nothing to do (a.Lq / a.Lk bound every loop and store).  Scalar (per-block uniform) arithmetic only.
__device__ __forceinline__ AttnArgs localize(const AttnArgs& a0, int b, int h) {
  AttnArgs a = a0;
  if (a0.cu_q != nullptr) {
    const int q0 = a0.cu_q[b], k0 = a0.cu_k[b];
    const int Lq = a0.cu_q[b + 1] - q0, Lk = a0.cu_k[b + 1] - k0;
    const long sq = (long)q0 - (long)b * Lq, sk = (long)k0 - (long)b * Lk;
    a.Q += sq * a.ldq;
    if (a.O != nullptr) a.O += sq * a.ldo;
    if (a.dO != nullptr) a.dO += sq * a.lddo;
    if (a.dQ != nullptr) a.dQ += sq * a.lddq;
    a.K += sk * a.ldk;
    a.V += sk * a.ldv;
    if (a.dK != nullptr) a.dK += sk * a.lddk;
    if (a.dV != nullptr) a.dV += sk * a.lddv;
    const long bh = (long)b * a.H + h;
    if (a.lse != nullptr) a.lse += bh * (a0.Lq - Lq);
    if (a.delta != nullptr) a.delta += bh * (a0.Lq - Lq);
    if (a.mbits != nullptr) a.mbits += bh * ((long)a0.Lq * ((a0.Lk + 63) >> 6) - (long)Lq * ((Lk + 63) >> 6));
    a.Lq = Lq; a.Lk = Lk;
    a.kpm = nullptr;
  }
  return a;
}

// v_exp_f32 directly: arguments are <= 0 (or -inf), no denormal-range fix-up needed
#define EXP2(x) __builtin_amdgcn_exp2f(x)
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

template <int HD> struct AttnGeom {
  static constexpr int HDP = (HD + 31) / 32 * 32;
  static constexpr int KS = HDP / 32;     // k-steps over the head dim
  static constexpr int DT = HD / 16;      // 16-wide output tiles over the head dim
  static constexpr int STRIDE = HDP * 2 + 32;
  static constexpr int CH = HD / 8;       // 16-byte chunks per row (data)
  static constexpr int CHP = HDP / 8;     // 16-byte chunks per row (incl. zero pad)
};

__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16_t)0.f;
  return z;
}

// transposed fragment: lane (g,i) gets T[rows row0+4g+{0..3} and row0+16+4g+{0..3}][col0+i]
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int stride, int row0, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const char* a0 = tile + (row0 + 4 * g + (i >> 2)) * stride + (col0 + 4 * (i & 3)) * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(a0 + 16 * stride));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int stride, int row, int chunk) {
  return *(LDS_PTR(const bf16x8))(tile + row * stride + chunk * 16);
}

// cooperative load of a [NR rows][HD] tile (rows >= nvalid and pad columns zero-filled)
template <int HD, int NR, int NT>
__device__ __forceinline__ void load_tile(char* tile, const bf16_t* base, long ld, int row_first, int nrows_total, int tid) {
  using G = AttnGeom<HD>;
  for (int id = tid; id < NR * G::CHP; id += NT) {
    const int r = id / G::CHP, c = id - r * G::CHP;
    bf16x8 v = zero8();
    if (row_first + r < nrows_total && c < G::CH) v = *(const bf16x8*)(base + (long)(row_first + r) * ld + c * 8);
    *(LDS_PTR(bf16x8))(tile + r * G::STRIDE + c * 16) = v;
  }
}

// Register-staged variant (T14 split): fetch = issue the global loads of a tile into registers,
// commit = write them to LDS after the barrier.  The fetch of tile t+1 is issued right after tile t has
// been committed, so its HBM/L2 latency hides under tile t's MFMA + softmax work.
template <int HD, int NR, int NT> struct TileRegs {
  static constexpr int NPT = (NR * AttnGeom<HD>::CHP + NT - 1) / NT;
  bf16x8 v[NPT];
};
// Raw-buffer loads: rows past the end of the sequence (and the padding chunks of the LDS image) carry an offset beyond
// num_records and come back as zeros from the hardware range check -- no EXEC-mask branches around the loads.
template <int HD, int NR, int NT>
__device__ __forceinline__ void tile_fetch(TileRegs<HD, NR, NT>& t, const bf16_t* base, long ld, int row_first, int nrows_total, int tid) {
  using G = AttnGeom<HD>;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)row_first * ld), 0, (int)0x80000000u, 0x00020000);
  const int nleft = nrows_total - row_first;
#pragma unroll
  for (int k = 0; k < TileRegs<HD, NR, NT>::NPT; ++k) {
    const int id = tid + k * NT;
    const int r = id / G::CHP, c = id - r * G::CHP;
    const bool ok = id < NR * G::CHP && r < nleft && c < G::CH;
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? (int)((r * ld + c * 8) * 2) : (int)0x80000000u, 0, 0);
    t.v[k] = __builtin_bit_cast(bf16x8, v);
  }
}
template <int HD, int NR, int NT>
__device__ __forceinline__ void tile_commit(const TileRegs<HD, NR, NT>& t, char* tile, int tid) {
  using G = AttnGeom<HD>;
#pragma unroll
  for (int k = 0; k < TileRegs<HD, NR, NT>::NPT; ++k) {
    const int id = tid + k * NT;
    const int r = id / G::CHP, c = id - r * G::CHP;
    if (id < NR * G::CHP) *(LDS_PTR(bf16x8))(tile + r * G::STRIDE + c * 16) = t.v[k];
  }
}

#ifndef DKV_QT
#define DKV_QT 32        // query rows per tile of the dK/dV kernel (64 was measured: +33 % time, the staging registers cost a wave per SIMD)
#endif

// XCD-aware block -> (tile, batch*head) map.  Blocks are dealt round-robin over the 8 XCDs (private L2s), so
// with the natural (tile fastest) order the tiles of one (batch, head) land on 8 different L2s and each
// re-fetches that head's K/V (or Q/dO) through the fabric: 230-390 MB per launch measured (profiles/
// r01_hbm_traffic.json).  Here blocks l, l+8, l+16, ... (one XCD) walk the tiles of the SAME (batch, head)
// before moving on, so its operands are fetched once per XCD.  Placement only affects speed.
__device__ __forceinline__ void tile_and_head(int ntiles, int nbh, int& tile, int& bh, int l = blockIdx.x) {
  const int full = (nbh >> 3) << 3;                 // heads covered by complete groups of 8
  const int lim = full * ntiles;
  if (l < lim) {
    const int r = l >> 3;
    tile = r % ntiles;
    bh = (r / ntiles) * 8 + (l & 7);
  } else {                                          // remainder heads: natural order
    const int r = l - lim;
    tile = r % ntiles;
    bh = full + r / ntiles;
  }
}

// ------------------------------------------------------------------------------------------ forward
template <int HD, int NW, int QW, bool WB>     // WB: also write the dropout keep-mask as bit words (AttnArgs::mbits)
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(const AttnArgs a0) {
  using G = AttnGeom<HD>;
  constexpr int KS = G::KS, DT = G::DT, STRIDE = G::STRIDE, NT = NW * 64;
  __shared__ __attribute__((aligned(16))) char lds[2 * 64 * STRIDE + 64 * 4 + 16];
  char* Kt = lds;
  char* Vt = lds + 64 * STRIDE;
  float* mb = (float*)(lds + 2 * 64 * STRIDE);
  int* padflag = (int*)(lds + 2 * 64 * STRIDE + 64 * 4);       // does the current key tile contain PAD keys?

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
  int tile, bh;
  tile_and_head((a0.Lq + NW * QW * 16 - 1) / (NW * QW * 16), a0.B * a0.H, tile, bh);
  const int b = bh / a0.H, h = bh - b * a0.H;
  const AttnArgs a = localize(a0, b, h);
  if (tile * (NW * QW * 16) >= a.Lq) return;          // packed sequences: this sample is shorter than the longest one
  const int qbase = tile * (NW * QW * 16) + wave * QW * 16;

  bf16x8 qf[QW][KS];
#pragma unroll
  for (int qs = 0; qs < QW; ++qs) {
    const int qc = min(qbase + qs * 16 + i, a.Lq - 1);
    const bf16_t* qp = a.Q + ((long)b * a.Lq + qc) * a.ldq + h * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int e = ks * 32 + 8 * g;
      qf[qs][ks] = (e < HD) ? *(const bf16x8*)(qp + e) : zero8();
    }
  }
  f32x4 o[QW][DT];
  float m[QW], l[QW];
#pragma unroll
  for (int qs = 0; qs < QW; ++qs) {
    m[qs] = -INFINITY; l[qs] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[qs][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float sl2 = a.scale * LOG2E;
  const uint32_t key32 = site_key(eff_seed(a.seed, a.seed_dev), a.site, (uint32_t)((a.b_offset + b) * a.H + h));
  const bf16_t* Kb = a.K + (long)b * a.Lk * a.ldk + h * HD;
  const bf16_t* Vb = a.V + (long)b * a.Lk * a.ldv + h * HD;
  const int nkt = (a.Lk + 63) >> 6;

  constexpr bool PF = true;               // register prefetch of the next K/V tile
  TileRegs<HD, 64, NT> kr, vr;
  if (PF) {
    tile_fetch<HD, 64, NT>(kr, Kb, a.ldk, 0, a.Lk, tid);
    tile_fetch<HD, 64, NT>(vr, Vb, a.ldv, 0, a.Lk, tid);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    if (PF) {
      tile_commit<HD, 64, NT>(kr, Kt, tid);
      tile_commit<HD, 64, NT>(vr, Vt, tid);
    } else {
      load_tile<HD, 64, NT>(Kt, Kb, a.ldk, kt * 64, a.Lk, tid);
      load_tile<HD, 64, NT>(Vt, Vb, a.ldv, kt * 64, a.Lk, tid);
    }
    if (tid < 64) {
      const int key = kt * 64 + tid;
      const bool pad = key >= a.Lk || (a.kpm != nullptr && a.kpm[(long)b * a.Lk + key] != 0);
      mb[tid] = pad ? -INFINITY : 0.f;
      const bool anyp = __builtin_amdgcn_ballot_w64(pad) != 0;
      if (tid == 0) padflag[0] = anyp ? 1 : 0;
    }
    __syncthreads();
    if (PF && kt + 1 < nkt) {
      tile_fetch<HD, 64, NT>(kr, Kb, a.ldk, (kt + 1) * 64, a.Lk, tid);
      tile_fetch<HD, 64, NT>(vr, Vb, a.ldv, (kt + 1) * 64, a.Lk, tid);
    }

    f32x4 s[QW][4];
#pragma unroll
    for (int qs = 0; qs < QW; ++qs)
#pragma unroll
      for (int n = 0; n < 4; ++n) s[qs][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = row_frag(Kt, STRIDE, n * 16 + i, ks * 4 + g);
#pragma unroll
        for (int qs = 0; qs < QW; ++qs) s[qs][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qs][ks], s[qs][n], 0, 0, 0);
      }

    const bool has_pad = padflag[0] != 0;          // block-uniform
#pragma unroll
    for (int qs = 0; qs < QW; ++qs) {
      // scores stay RAW (unscaled) until the exponent: scale > 0 commutes with the maximum, and
      // p = exp2(s*sl2 - m) is then one fma + one exp per element.  Only tiles with PAD keys add the -inf bias.
      float mx = -INFINITY;
      if (has_pad) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const f32x4 bias = *(LDS_PTR(const f32x4))(mb + n * 16 + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) s[qs][n][r] += bias[r];
        }
      }
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qs][n][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      mx *= sl2;                                     // -inf stays -inf
      // Deferred rescale: m[qs] is the reference exponent the running sums are expressed in, not the running
      // maximum.  It only moves (and O, l are only rescaled) when some query of the wave has grown past it by more
      // than 8 in log2 units, so p stays <= 2^8 and most key tiles skip the 48-register rescale of O altogether.
      const float mnew = fmaxf(m[qs], mx);
      if (__builtin_amdgcn_ballot_w64(mnew > m[qs] + 8.f) != 0) {     // wave-uniform; -inf + 8 = -inf: first valid tile
        const float ms = (mnew == -INFINITY) ? 0.f : mnew;
        const float alpha = EXP2(m[qs] - ms);
        m[qs] = mnew;
        l[qs] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qs][dt] *= alpha;
      }
      const float nm = (m[qs] == -INFINITY) ? 0.f : -m[qs];
      float rs = 0.f;
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = EXP2(fmaf(s[qs][n][r], sl2, nm));
          s[qs][n][r] = p;
          rs += p;
        }
      rs += __shfl_xor(rs, 16);
      rs += __shfl_xor(rs, 32);
      l[qs] += rs;
      if (a.thr16 != 0) {
        // keys 4g..4g+3 of each 16-key subtile = two hash pairs; one multiply-free step per extra pair.  The
        // 1/(1-p) factor of the kept probabilities is applied once, to O, after the last key tile.
        const uint32_t hb = drop_base(key32, (uint32_t)(qbase + qs * 16 + i), (uint32_t)((kt * 64 + 4 * g) >> 1));
        uint32_t bits = 0u;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const uint32_t x = mix24(hb + (uint32_t)(n * 8 + pr) * DROP_CB);
            const bool k0 = keep_lo(x, a.thr16), k1 = keep_hi(x, a.thr16);
            s[qs][n][2 * pr] = k0 ? s[qs][n][2 * pr] : 0.f;
            s[qs][n][2 * pr + 1] = k1 ? s[qs][n][2 * pr + 1] : 0.f;
            if (WB) bits |= (k0 ? 1u : 0u) << (n * 4 + 2 * pr) | (k1 ? 1u : 0u) << (n * 4 + 2 * pr + 1);
          }
        if (WB && qbase + qs * 16 + i < a.Lq)       // this lane's 16 bits of the (query, key tile) word
          ((unsigned short*)(a.mbits + (((long)bh * a.Lq + qbase + qs * 16 + i) * nkt + kt)))[g] = (unsigned short)bits;
      }
    }

#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 pf[QW];
#pragma unroll
      for (int qs = 0; qs < QW; ++qs)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[qs][j] = (bf16_t)s[qs][2 * s2 + (j >> 2)][j & 3];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bf16x8 vf = tr_frag(Vt, STRIDE, 32 * s2, dt * 16, lane);
#pragma unroll
        for (int qs = 0; qs < QW; ++qs) o[qs][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qs], o[qs][dt], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int qs = 0; qs < QW; ++qs) {
    const int q = qbase + qs * 16 + i;
    if (q >= a.Lq) continue;
    const float inv = (a.thr16 != 0 ? a.inv_keep : 1.f) / l[qs];   // l == 0 (all keys PAD) -> 0 * inf = NaN, as the reference
    bf16_t* op = a.O + ((long)b * a.Lq + q) * a.ldo + h * HD + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      bf16x4 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = (bf16_t)(o[qs][dt][r] * inv);
      *(bf16x4*)(op + dt * 16) = w;
    }
    if (g == 0 && a.lse != nullptr) a.lse[((long)b * a.H + h) * a.Lq + q] = (m[qs] + log2f(l[qs])) * LN2;
  }
  if constexpr (DT % 2 == 0) {
    if (a.Oq != nullptr) {            // kernel-uniform (fp8 GEMM mode): the MX-fp8 form of the stored O for the out-projection
      // a 32-column MX block of row q = sub-tiles dt = 2k, 2k+1 over the four lane groups g of lane column i: 8 values per lane,
      // block maximum over lanes i, i+16, i+32, i+48; quantises the ROUNDED bf16 values (bit-identical to hriemo_quant_mx8 of O)
#pragma unroll
      for (int qs = 0; qs < QW; ++qs) {
        const int q = qbase + qs * 16 + i;
        const float inv = (a.thr16 != 0 ? a.inv_keep : 1.f) / l[qs];
        const long row = (long)b * a.Lq + min(q, a.Lq - 1);
#pragma unroll
        for (int kb = 0; kb < DT / 2; ++kb) {
          float f[8];
          float amax = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            f[j] = (float)(bf16_t)(o[qs][2 * kb + (j >> 2)][j & 3] * inv);
            amax = fmaxf(amax, fabsf(f[j]));
          }
          amax = fmaxf(amax, __shfl_xor(amax, 16));
          amax = fmaxf(amax, __shfl_xor(amax, 32));
          const unsigned ab = __float_as_uint(amax);
          int e = (int)(ab >> 23) - 8 + ((ab & 0x7fffffu) > 0x600000u ? 1 : 0);
          e = amax == 0.f ? 0 : (e < 1 ? 1 : (e > 253 ? 253 : e));
          const float sc = amax == 0.f ? 0.f : __uint_as_float((unsigned)(254 - e) << 23);
          int w0 = 0, w1 = 0;
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * sc, f[1] * sc, w0, false);
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * sc, f[3] * sc, w0, true);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * sc, f[5] * sc, w1, false);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * sc, f[7] * sc, w1, true);
          if (q < a.Lq) {
            uint8_t* oq = a.Oq + row * a.ldoq + h * HD + kb * 32 + 4 * g;
            *(int*)oq = w0;
            *(int*)(oq + 16) = w1;
            if (g == 0) a.So[(long)((h * HD) / 32 + kb) * a.ldso + row] = (uint8_t)e;
          }
        }
      }
    }
  }
}

// Column sums of a block's output tile (bias gradients of the packed in-projection): a lane holds columns
// dt*16 + 4g .. +3 of row i of every 16-row sub-tile.  Sum over the 16 lanes of a row group, then over the waves through
// LDS (free by now: the caller has passed a barrier after its last tile), one fp32 row of HD columns per block.
__device__ __forceinline__ int gridDim_tiles(int L, int rows_per_block) { return (L + rows_per_block - 1) / rows_per_block; }
template <int DT, int NW>
__device__ __forceinline__ void block_colsum_store(f32x4 (&cs)[DT], float* red, float* dst, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    f32x4 v = cs[dt];
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], m, 64);
    }
    if ((lane & 15) == 0) *(f32x4*)(red + wave * (DT * 16) + dt * 16 + 4 * (lane >> 4)) = v;
  }
  __syncthreads();
  for (int c = tid; c < DT * 16; c += NW * 64) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += red[w * (DT * 16) + c];
    dst[c] = t;
  }
}

// ------------------------------------------------------------------------------------------ dQ (+ delta)
template <int HD, int NW, int QW, bool BITS>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(const AttnArgs a0) {
  using G = AttnGeom<HD>;
  constexpr int KS = G::KS, DT = G::DT, STRIDE = G::STRIDE, NT = NW * 64;
  __shared__ __attribute__((aligned(16))) char lds[2 * 64 * STRIDE + 64 * 4];
  char* Kt = lds;
  char* Vt = lds + 64 * STRIDE;
  float* mb = (float*)(lds + 2 * 64 * STRIDE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
  int tile, bh;
  tile_and_head((a0.Lq + NW * QW * 16 - 1) / (NW * QW * 16), a0.B * a0.H, tile, bh);
  const int b = bh / a0.H, h = bh - b * a0.H;
  const AttnArgs a = localize(a0, b, h);
  float* const csq_row = a.csq != nullptr ? a.csq + (long)(b * gridDim_tiles(a0.Lq, NW * QW * 16) + tile) * ((long)a.H * HD) + h * HD : nullptr;
  if (tile * (NW * QW * 16) >= a.Lq) {                 // packed sequences: no query of this sample in the tile
    if (csq_row != nullptr) for (int e = tid; e < HD; e += NT) csq_row[e] = 0.f;
    return;
  }
  const int qbase = tile * (NW * QW * 16) + wave * QW * 16;

  bf16x8 qf[QW][KS], dof[QW][KS];
  float lse2[QW], dl[QW];
#pragma unroll
  for (int qs = 0; qs < QW; ++qs) {
    const int q = qbase + qs * 16 + i;
    const int qc = min(q, a.Lq - 1);
    const bf16_t* qp = a.Q + ((long)b * a.Lq + qc) * a.ldq + h * HD;
    const bf16_t* dop = a.dO + ((long)b * a.Lq + qc) * a.lddo + h * HD;
    const bf16_t* op = a.O + ((long)b * a.Lq + qc) * a.ldo + h * HD;
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int e = ks * 32 + 8 * g;
      if (e < HD) {
        qf[qs][ks] = *(const bf16x8*)(qp + e);
        dof[qs][ks] = *(const bf16x8*)(dop + e);
        const bf16x8 ov = *(const bf16x8*)(op + e);
#pragma unroll
        for (int j = 0; j < 8; ++j) part += (float)dof[qs][ks][j] * (float)ov[j];
      } else {
        qf[qs][ks] = zero8();
        dof[qs][ks] = zero8();
      }
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    dl[qs] = part;
    const long li = ((long)b * a.H + h) * a.Lq + qc;
    lse2[qs] = a.lse[li] * LOG2E;
    if (g == 0 && q < a.Lq) a.delta[li] = part;
  }
  f32x4 dq[QW][DT];
#pragma unroll
  for (int qs = 0; qs < QW; ++qs)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[qs][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float sl2 = a.scale * LOG2E;
  float cexp[QW], dlk[QW];
#pragma unroll
  for (int qs = 0; qs < QW; ++qs) {
    cexp[qs] = (a.thr16 != 0 ? log2f(a.inv_keep) : 0.f) - lse2[qs];
    dlk[qs] = a.thr16 != 0 ? dl[qs] / a.inv_keep : dl[qs];
  }
  const uint32_t key32 = site_key(eff_seed(a.seed, a.seed_dev), a.site, (uint32_t)((a.b_offset + b) * a.H + h));
  const bf16_t* Kb = a.K + (long)b * a.Lk * a.ldk + h * HD;
  const bf16_t* Vb = a.V + (long)b * a.Lk * a.ldv + h * HD;
  const int nkt = (a.Lk + 63) >> 6;

  constexpr bool PF = (NW == 4);
  TileRegs<HD, 64, NT> kr, vr;
  if (PF) {
    tile_fetch<HD, 64, NT>(kr, Kb, a.ldk, 0, a.Lk, tid);
    tile_fetch<HD, 64, NT>(vr, Vb, a.ldv, 0, a.Lk, tid);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    if (PF) {
      tile_commit<HD, 64, NT>(kr, Kt, tid);
      tile_commit<HD, 64, NT>(vr, Vt, tid);
    } else {
      load_tile<HD, 64, NT>(Kt, Kb, a.ldk, kt * 64, a.Lk, tid);
      load_tile<HD, 64, NT>(Vt, Vb, a.ldv, kt * 64, a.Lk, tid);
    }
    if (tid < 64) {
      const int key = kt * 64 + tid;
      const bool pad = key >= a.Lk || (a.kpm != nullptr && a.kpm[(long)b * a.Lk + key] != 0);
      mb[tid] = pad ? -INFINITY : 0.f;
    }
    __syncthreads();
    if (PF && kt + 1 < nkt) {
      tile_fetch<HD, 64, NT>(kr, Kb, a.ldk, (kt + 1) * 64, a.Lk, tid);
      tile_fetch<HD, 64, NT>(vr, Vb, a.ldv, (kt + 1) * 64, a.Lk, tid);
    }

    f32x4 s[QW][4], dp[QW][4];
#pragma unroll
    for (int qs = 0; qs < QW; ++qs)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        s[qs][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dp[qs][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = row_frag(Kt, STRIDE, n * 16 + i, ks * 4 + g);
        const bf16x8 vf = row_frag(Vt, STRIDE, n * 16 + i, ks * 4 + g);
#pragma unroll
        for (int qs = 0; qs < QW; ++qs) {
          s[qs][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qs][ks], s[qs][n], 0, 0, 0);
          dp[qs][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[qs][ks], dp[qs][n], 0, 0, 0);
        }
      }
#pragma unroll
    for (int qs = 0; qs < QW; ++qs) {
      const uint32_t hb = drop_base(key32, (uint32_t)(qbase + qs * 16 + i), (uint32_t)((kt * 64 + 4 * g) >> 1));
      uint32_t bits = 0xffffu;                            // dropout off: everything kept
      constexpr bool from_bits = BITS;
      if (from_bits && a.thr16 != 0)                      // same register order as the forward: this lane's own 16 bits
        bits = ((const unsigned short*)(a.mbits + (((long)bh * a.Lq + min(qbase + qs * 16 + i, a.Lq - 1)) * nkt + kt)))[g];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const f32x4 bias = *(LDS_PTR(const f32x4))(mb + n * 16 + 4 * g);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          uint32_t x = 0xffffffffu;                       // dropout off: both halves >= any threshold
          if (a.thr16 != 0 && !from_bits) x = mix24(hb + (uint32_t)(n * 8 + pr) * DROP_CB);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int r = 2 * pr + e;
            // pk = p / (1 - p_drop): the dropout scale rides in the exponent's constant, delta is pre-scaled by
            // (1 - p_drop), so dS = pk * (keep ? dP : 0  -  delta') is one fma + exp + select + sub + mul
            const float pk = EXP2(fmaf(s[qs][n][r], sl2, bias[r] + cexp[qs]));
            float dpd = dp[qs][n][r];
            if (a.thr16 != 0) {
              const bool keep = from_bits ? ((bits >> (n * 4 + r)) & 1u) != 0u : (e ? keep_hi(x, a.thr16) : keep_lo(x, a.thr16));
              dpd = keep ? dpd : 0.f;
            }
            s[qs][n][r] = pk * (dpd - dlk[qs]);
          }
        }
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 dsf[QW];
#pragma unroll
      for (int qs = 0; qs < QW; ++qs)
#pragma unroll
        for (int j = 0; j < 8; ++j) dsf[qs][j] = (bf16_t)s[qs][2 * s2 + (j >> 2)][j & 3];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bf16x8 ktf = tr_frag(Kt, STRIDE, 32 * s2, dt * 16, lane);
#pragma unroll
        for (int qs = 0; qs < QW; ++qs) dq[qs][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[qs], dq[qs][dt], 0, 0, 0);
      }
    }
  }
  f32x4 cs[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) cs[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int qs = 0; qs < QW; ++qs) {
    const int q = qbase + qs * 16 + i;
    if (q >= a.Lq) continue;
    bf16_t* dqp = a.dQ + ((long)b * a.Lq + q) * a.lddq + h * HD + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      bf16x4 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = dq[qs][dt][r] * a.scale;
        w[r] = (bf16_t)v;
        cs[dt][r] += v;                           // fp32 values BEFORE the bf16 rounding: the bias gradient does not inherit it
      }
      *(bf16x4*)(dqp + dt * 16) = w;
    }
  }
  if (a.csq != nullptr) {          // kernel-uniform
    __syncthreads();                // every wave is done with the K/V tiles
    block_colsum_store<DT, NW>(cs, (float*)lds, csq_row, tid);
  }
}

// ------------------------------------------------------------------------------------------ dQ, dK, dV: query-resident single pass
// The mirror of the key-resident single-pass kernel further down, for L_q <= 128 < L_k (text queries over audio keys, t2a at
// cfg 2: L_q = 128, L_k = 400; the MOSEI shape's 50 x 1000): ONE 512-thread block holds ALL queries of a (batch, head) -- each of
// its 8 waves 16 of them, on the lanes (S^T = K.Q^T as in the dQ kernel above) -- and sweeps the keys in 64-row tiles:
//   S^T, dP^T          24 MFMAs per wave and tile; softmax / dropout arithmetic of the dQ kernel
//   dQ^T += K^T.dS^T   12 MFMAs, dS^T straight from the accumulators as the B operand; dQ stays in registers over the sweep
//   P~ and dS cross LDS once ([query][key] bf16 images, 8-byte stores) and every wave multiplies one 16-key slice of
//   dV^T = dO^T.P~ (even waves) or dK^T = Q^T.dS (odd waves) over all 128 queries (24 MFMAs): these tiles are COMPLETE -- the block
//   holds every query -- and are stored at once; Q and dO stay resident as LDS images for those products.
// 5 GEMMs per tile instead of the two-kernel path's 7, Q / K / V / dO read once, no delta round trip, no atomics, deterministic.
// Two barriers per tile: (C) P~ / dS images complete + K / V tile no longer read, (B) next K / V tile committed + images free.
template <int HD, bool BITS>
__global__ __launch_bounds__(512) void attn_bwd_qres_kernel(const AttnArgs a0) {
  using G = AttnGeom<HD>;
  constexpr int KS = G::KS, DT = G::DT, STRIDE = G::STRIDE, NW = 8, NT = 512, NQ = 128, KT = 64;
  constexpr int PS = KT * 2 + 32;                    // row stride of the P~ / dS images ([query][64 keys] bf16 + pad)
  __shared__ __attribute__((aligned(16))) char lds[2 * NQ * STRIDE + 2 * KT * STRIDE + 2 * NQ * PS + KT * 4];
  char* const Qi = lds;
  char* const dOi = Qi + NQ * STRIDE;
  char* const Kt = dOi + NQ * STRIDE;
  char* const Vt = Kt + KT * STRIDE;
  char* const Pi = Vt + KT * STRIDE;
  char* const dSi = Pi + NQ * PS;
  float* const mb = (float*)(dSi + NQ * PS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
  const int bh = blockIdx.x;
  const int b = bh / a0.H, h = bh - b * a0.H;
  const AttnArgs a = localize(a0, b, h);
  const bf16_t* Qb = a.Q + (long)b * a.Lq * a.ldq + h * HD;
  const bf16_t* dOb = a.dO + (long)b * a.Lq * a.lddo + h * HD;
  const bf16_t* Kb = a.K + (long)b * a.Lk * a.ldk + h * HD;
  const bf16_t* Vb = a.V + (long)b * a.Lk * a.ldv + h * HD;
  const int nkt = (a.Lk + KT - 1) / KT;
  TileRegs<HD, KT, NT> kr, vr;
  tile_fetch<HD, KT, NT>(kr, Kb, a.ldk, 0, a.Lk, tid);
  tile_fetch<HD, KT, NT>(vr, Vb, a.ldv, 0, a.Lk, tid);
  load_tile<HD, NQ, NT>(Qi, Qb, a.ldq, 0, a.Lq, tid);           // rows >= L_q zero
  load_tile<HD, NQ, NT>(dOi, dOb, a.lddo, 0, a.Lq, tid);

  // this lane's query (wave * 16 + i): delta = rowsum(dO * O) over the head dim (lane groups g share the work), lse
  const int q = wave * 16 + i;
  const bool qok = q < a.Lq;
  const int qc = min(q, a.Lq - 1);
  float part = 0.f;
  {
    const bf16_t* dop = dOb + (long)qc * a.lddo;
    const bf16_t* op = a.O + ((long)b * a.Lq + qc) * a.ldo + h * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int e = ks * 32 + 8 * g;
      if (e < HD) {
        const bf16x8 dv8 = *(const bf16x8*)(dop + e), ov = *(const bf16x8*)(op + e);
#pragma unroll
        for (int j = 0; j < 8; ++j) part += (float)dv8[j] * (float)ov[j];
      }
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
  }
  const long li = ((long)b * a.H + h) * a.Lq + qc;
  // rows past L_q: exponent constant -inf -> P~ = dS = 0, they add nothing to dK / dV
  const float cexp = qok ? (a.thr16 != 0 ? log2f(a.inv_keep) : 0.f) - a.lse[li] * LOG2E : -INFINITY;
  const float dlk = a.thr16 != 0 ? part / a.inv_keep : part;
  const float sl2 = a.scale * LOG2E;
  const uint32_t key32 = site_key(eff_seed(a.seed, a.seed_dev), a.site, (uint32_t)((a.b_offset + b) * a.H + h));

  tile_commit<HD, KT, NT>(kr, Kt, tid);
  tile_commit<HD, KT, NT>(vr, Vt, tid);
  if (tid < KT) {
    const bool pad = tid >= a.Lk || (a.kpm != nullptr && a.kpm[(long)b * a.Lk + tid] != 0);
    mb[tid] = pad ? -INFINITY : 0.f;
  }
  __syncthreads();
  bf16x8 qf[KS], dof[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = row_frag(Qi, STRIDE, q, ks * 4 + g);
    dof[ks] = row_frag(dOi, STRIDE, q, ks * 4 + g);
  }
  f32x4 dq[DT], cs[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; cs[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const int which = wave & 1, nsub = wave >> 1;       // dV (even waves) / dK (odd waves) of key sub-tile nsub of every tile
  const char* const Yimg = which ? dSi : Pi;
  const char* const Ximg = which ? Qi : dOi;
  bf16_t* const dOut = which ? a.dK : a.dV;
  const long ldout = which ? a.lddk : a.lddv;
  const float oscale = which ? a.scale : 1.f;

  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) {
      tile_fetch<HD, KT, NT>(kr, Kb, a.ldk, (kt + 1) * KT, a.Lk, tid);
      tile_fetch<HD, KT, NT>(vr, Vb, a.ldv, (kt + 1) * KT, a.Lk, tid);
    }
    f32x4 s[4], dp[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) { s[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = row_frag(Kt, STRIDE, n * 16 + i, ks * 4 + g);
        const bf16x8 vf = row_frag(Vt, STRIDE, n * 16 + i, ks * 4 + g);
        s[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[n], 0, 0, 0);
        dp[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[ks], dp[n], 0, 0, 0);
      }
    {
      const uint32_t hb = drop_base(key32, (uint32_t)q, (uint32_t)((kt * KT + 4 * g) >> 1));
      uint32_t bits = 0xffffu;
      if (BITS && a.thr16 != 0) bits = ((const unsigned short*)(a.mbits + (((long)bh * a.Lq + qc) * nkt + kt)))[g];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const f32x4 bias = *(LDS_PTR(const f32x4))(mb + n * 16 + 4 * g);
        bf16x4 pw, dw;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          uint32_t x = 0xffffffffu;
          if (a.thr16 != 0 && !BITS) x = mix24(hb + (uint32_t)(n * 8 + pr) * DROP_CB);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int r = 2 * pr + e;
            const float pk = EXP2(fmaf(s[n][r], sl2, bias[r] + cexp));
            float dpd = dp[n][r], pd = pk;
            if (a.thr16 != 0) {
              const bool keep = BITS ? ((bits >> (n * 4 + r)) & 1u) != 0u : (e ? keep_hi(x, a.thr16) : keep_lo(x, a.thr16));
              dpd = keep ? dpd : 0.f;
              pd = keep ? pk : 0.f;
            }
            const float dsv = pk * (dpd - dlk);
            s[n][r] = dsv;
            pw[r] = (bf16_t)pd;
            dw[r] = (bf16_t)dsv;
          }
        }
        // P~[query][keys 16n + 4g .. +3] and dS likewise: one 8-byte store each
        *(LDS_PTR(bf16x4))(Pi + q * PS + (n * 16 + 4 * g) * 2) = pw;
        *(LDS_PTR(bf16x4))(dSi + q * PS + (n * 16 + 4 * g) * 2) = dw;
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 dsf;
#pragma unroll
      for (int j = 0; j < 8; ++j) dsf[j] = (bf16_t)s[2 * s2 + (j >> 2)][j & 3];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bf16x8 ktf = tr_frag(Kt, STRIDE, 32 * s2, dt * 16, lane);
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, dq[dt], 0, 0, 0);
      }
    }
    __syncthreads();          // (C) every wave's P~ / dS rows are in LDS; nobody reads this K / V tile or its mask any more
    if (kt + 1 < nkt) {
      tile_commit<HD, KT, NT>(kr, Kt, tid);
      tile_commit<HD, KT, NT>(vr, Vt, tid);
      if (tid < KT) {
        const int key = (kt + 1) * KT + tid;
        const bool pad = key >= a.Lk || (a.kpm != nullptr && a.kpm[(long)b * a.Lk + key] != 0);
        mb[tid] = pad ? -INFINITY : 0.f;
      }
    }
    {
      // out^T[head dim][16 keys] = X^T . Y over all 128 queries: X = dO (dV) or Q (dK), Y = P~ or dS, both read transposed
      // in tr_frag's permuted k-order
      f32x4 acc[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < NQ / 32; ++kk) {
        const bf16x8 yf = tr_frag(Yimg, PS, 32 * kk, nsub * 16, lane);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const bf16x8 xf = tr_frag(Ximg, STRIDE, 32 * kk, dt * 16, lane);
          acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, yf, acc[dt], 0, 0, 0);
        }
      }
      const int key = kt * KT + nsub * 16 + i;
      if (key < a.Lk) {
        bf16_t* op = dOut + ((long)b * a.Lk + key) * ldout + h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          bf16x4 w;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[dt][r] * oscale;
            w[r] = (bf16_t)v;
            cs[dt][r] += v;
          }
          *(bf16x4*)(op + dt * 16) = w;
        }
      }
    }
    __syncthreads();          // (B) next K / V tile and mask are in place; the P~ / dS images may be overwritten
  }
  // dQ rows of this wave + column sums (in-projection bias gradients): dQ per block, dK | dV per block (the block holds all keys)
  f32x4 csq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) csq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (qok) {
    bf16_t* dqp = a.dQ + ((long)b * a.Lq + q) * a.lddq + h * HD + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      bf16x4 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = dq[dt][r] * a.scale;
        w[r] = (bf16_t)v;
        csq[dt][r] += v;
      }
      *(bf16x4*)(dqp + dt * 16) = w;
    }
  }
  float* red = (float*)lds;                     // [NW][DT * 16] floats: the images are dead behind the loop's last barrier
  if (a.csq != nullptr) {
    block_colsum_store<DT, NW>(csq, red, a.csq + (long)b * ((long)a.H * HD) + h * HD, tid);
    __syncthreads();
  }
  if (a.cskv != nullptr) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      f32x4 v = cs[dt];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], m, 64);
      }
      if (i == 0) *(f32x4*)(red + wave * (DT * 16) + dt * 16 + 4 * g) = v;
    }
    __syncthreads();
    float* row = a.cskv + (long)b * (2L * a.H * HD) + h * HD;
    for (int c = tid; c < 2 * DT * 16; c += NT) {
      const int wh = c / (DT * 16), e = c - wh * (DT * 16);          // wh 0: dK sums (odd waves), 1: dV sums (even waves)
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += red[(2 * w + (wh == 0 ? 1 : 0)) * (DT * 16) + e];
      row[(long)wh * a.H * HD + e] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------ dK, dV (two-kernel path, hash mask)
// The round-1 kernel, kept as it was tuned: the default backward (hash replay, dQ from its own kernel).  The kernel after it adds
// the bit-word mask and the fused dQ; its different tile pipeline costs this one's register budget (3 blocks per CU at head_dim 96).
template <int HD, int NW, int KW, int QT>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_hash_kernel(const AttnArgs a0) {
  using G = AttnGeom<HD>;
  constexpr int KS = G::KS, DT = G::DT, STRIDE = G::STRIDE, NT = NW * 64;
  static_assert(QT == 32 || QT == 64, "query tile");
  constexpr int NQS = QT / 16, NKQ = QT / 32;
  // two LDS images of the (Q, dO, lse, delta) tile: the next tile is committed while the current one is consumed,
  // so a query tile costs ONE barrier instead of two and the LDS stores overlap the MFMAs
  constexpr int TILE_BYTES = 2 * QT * STRIDE + 2 * QT * 4;
  __shared__ __attribute__((aligned(16))) char lds[2 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
  int tile, bh;
  tile_and_head((a0.Lk + NW * KW * 16 - 1) / (NW * KW * 16), a0.B * a0.H, tile, bh);
  const int b = bh / a0.H, h = bh - b * a0.H;
  const AttnArgs a = localize(a0, b, h);
  float* const cskv_row = a.cskv != nullptr ? a.cskv + (long)(b * gridDim_tiles(a0.Lk, NW * KW * 16) + tile) * (2L * a.H * HD) + h * HD : nullptr;
  if (tile * (NW * KW * 16) >= a.Lk) {                 // packed sequences: no key of this sample in the tile
    if (cskv_row != nullptr) for (int e = tid; e < HD; e += NT) { cskv_row[e] = 0.f; cskv_row[(long)a.H * HD + e] = 0.f; }
    return;
  }
  const int kbase = tile * (NW * KW * 16) + wave * KW * 16;

  bf16x8 kreg[KW][KS], vreg[KW][KS];
  bool kvalid[KW];
#pragma unroll
  for (int kw = 0; kw < KW; ++kw) {
    const int key = kbase + kw * 16 + i;
    const int kc = min(key, a.Lk - 1);
    kvalid[kw] = key < a.Lk && !(a.kpm != nullptr && a.kpm[(long)b * a.Lk + kc] != 0);
    const bf16_t* kp = a.K + ((long)b * a.Lk + kc) * a.ldk + h * HD;
    const bf16_t* vp = a.V + ((long)b * a.Lk + kc) * a.ldv + h * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int e = ks * 32 + 8 * g;
      kreg[kw][ks] = (e < HD) ? *(const bf16x8*)(kp + e) : zero8();
      vreg[kw][ks] = (e < HD) ? *(const bf16x8*)(vp + e) : zero8();
    }
  }
  uint32_t hkb[KW];                     // hash base of this lane's key: key32 + (key>>1)*CB  (a-term added per query)
#pragma unroll
  for (int kw = 0; kw < KW; ++kw) hkb[kw] = 0u;
  f32x4 dk[KW][DT], dv[KW][DT];
#pragma unroll
  for (int kw = 0; kw < KW; ++kw)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      dk[kw][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dv[kw][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  const float sl2 = a.scale * LOG2E;
  const uint32_t key32 = site_key(eff_seed(a.seed, a.seed_dev), a.site, (uint32_t)((a.b_offset + b) * a.H + h));
#pragma unroll
  for (int kw = 0; kw < KW; ++kw) hkb[kw] = drop_base(key32, 0u, (uint32_t)((kbase + kw * 16 + i) >> 1));
  const bf16_t* Qb = a.Q + (long)b * a.Lq * a.ldq + h * HD;
  const bf16_t* dOb = a.dO + (long)b * a.Lq * a.lddo + h * HD;
  const long lbase = ((long)b * a.H + h) * a.Lq;
  const int nqt = (a.Lq + QT - 1) / QT;

  const float l2ik = a.thr16 != 0 ? log2f(a.inv_keep) : 0.f, keepfrac = a.thr16 != 0 ? 1.f / a.inv_keep : 1.f;
  constexpr bool PF = (NW == 4);
  TileRegs<HD, QT, NT> qr, dor;
  float lse_r = 0.f, del_r = 0.f;                 // lse / delta of query row `tid` of the tile in flight (tid < QT)
  auto fetch = [&](int qt) {
    tile_fetch<HD, QT, NT>(qr, Qb, a.ldq, qt * QT, a.Lq, tid);
    tile_fetch<HD, QT, NT>(dor, dOb, a.lddo, qt * QT, a.Lq, tid);
    if (tid < QT) {
      const int q = qt * QT + tid;
      lse_r = q < a.Lq ? l2ik - a.lse[lbase + q] * LOG2E : -INFINITY;   // -inf -> p = 0 for rows past Lq
      del_r = q < a.Lq ? a.delta[lbase + q] * keepfrac : 0.f;
    }
  };
  auto commit = [&](int buf) {
    char* base = lds + buf * TILE_BYTES;
    tile_commit<HD, QT, NT>(qr, base, tid);
    tile_commit<HD, QT, NT>(dor, base + QT * STRIDE, tid);
    if (tid < QT) {
      ((float*)(base + 2 * QT * STRIDE))[tid] = lse_r;
      ((float*)(base + 2 * QT * STRIDE))[QT + tid] = del_r;
    }
  };
  fetch(0);
  commit(0);
  if (nqt > 1) fetch(1);
  __syncthreads();
  for (int qt = 0; qt < nqt; ++qt) {
    const char* Qt = lds + (qt & 1) * TILE_BYTES;
    const char* dOt = Qt + QT * STRIDE;
    const float* lse_s = (const float*)(Qt + 2 * QT * STRIDE);
    const float* del_s = lse_s + QT;
    if (qt + 1 < nqt) {
      commit((qt + 1) & 1);                       // image last read in iteration qt-1, released by its closing barrier
      if (qt + 2 < nqt) fetch(qt + 2);
    }

    f32x4 s[KW][NQS], dp[KW][NQS];
#pragma unroll
    for (int kw = 0; kw < KW; ++kw)
#pragma unroll
      for (int qs = 0; qs < NQS; ++qs) {
        s[kw][qs] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dp[kw][qs] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int qs = 0; qs < NQS; ++qs)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 qfr = row_frag(Qt, STRIDE, qs * 16 + i, ks * 4 + g);
        const bf16x8 dofr = row_frag(dOt, STRIDE, qs * 16 + i, ks * 4 + g);
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          s[kw][qs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kreg[kw][ks], s[kw][qs], 0, 0, 0);
          dp[kw][qs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dofr, vreg[kw][ks], dp[kw][qs], 0, 0, 0);
        }
      }
    bf16x8 pf[KW][NKQ], dsf[KW][NKQ];
#pragma unroll
    for (int qs = 0; qs < NQS; ++qs) {
      // lse_s holds (log2(1/(1-p_drop)) - lse*log2e), del_s holds delta*(1-p_drop): pk = p/(1-p_drop) straight
      // from the exponent, P~ = keep ? pk : 0, dS = pk * (keep ? dP : 0  -  delta')
      const f32x4 lse4 = *(LDS_PTR(const f32x4))(lse_s + qs * 16 + 4 * g);
      const f32x4 del4 = *(LDS_PTR(const f32x4))(del_s + qs * 16 + 4 * g);
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        const uint32_t key = (uint32_t)(kbase + kw * 16 + i);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pk = kvalid[kw] ? EXP2(fmaf(s[kw][qs][r], sl2, lse4[r])) : 0.f;
          float pd = pk, dpd = dp[kw][qs][r];
          if (a.thr16 != 0) {
            // the lane owns ONE key (pair index key>>1, half key&1) and walks the queries: a-term by addition
            const uint32_t x = mix24(hkb[kw] + (uint32_t)(qt * QT + qs * 16 + 4 * g + r) * DROP_CA);
            const bool keep = (key & 1u) ? keep_hi(x, a.thr16) : keep_lo(x, a.thr16);
            pd = keep ? pk : 0.f;
            dpd = keep ? dpd : 0.f;
          }
          pf[kw][qs >> 1][(qs & 1) * 4 + r] = (bf16_t)pd;
          dsf[kw][qs >> 1][(qs & 1) * 4 + r] = (bf16_t)(pk * (dpd - del4[r]));
        }
      }
    }
#pragma unroll
    for (int kq = 0; kq < NKQ; ++kq)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bf16x8 dotf = tr_frag(dOt, STRIDE, 32 * kq, dt * 16, lane);
        const bf16x8 qtf = tr_frag(Qt, STRIDE, 32 * kq, dt * 16, lane);
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          dv[kw][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dotf, pf[kw][kq], dv[kw][dt], 0, 0, 0);
          dk[kw][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf[kw][kq], dk[kw][dt], 0, 0, 0);
        }
      }
    __syncthreads();
  }
  f32x4 csk[DT], csv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { csk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; csv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int kw = 0; kw < KW; ++kw) {
    const int key = kbase + kw * 16 + i;
    if (key >= a.Lk) continue;
    bf16_t* dkp = a.dK + ((long)b * a.Lk + key) * a.lddk + h * HD + 4 * g;
    bf16_t* dvp = a.dV + ((long)b * a.Lk + key) * a.lddv + h * HD + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      bf16x4 wk, wv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float vk = dk[kw][dt][r] * a.scale, vv = dv[kw][dt][r];
        wk[r] = (bf16_t)vk;
        wv[r] = (bf16_t)vv;
        csk[dt][r] += vk;                         // fp32 values before the bf16 rounding
        csv[dt][r] += vv;
      }
      *(bf16x4*)(dkp + dt * 16) = wk;
      *(bf16x4*)(dvp + dt * 16) = wv;
    }
  }
  if (a.cskv != nullptr) {         // kernel-uniform
    float* row = cskv_row;
    __syncthreads();
    block_colsum_store<DT, NW>(csk, (float*)lds, row, tid);
    block_colsum_store<DT, NW>(csv, (float*)lds + NW * DT * 16, row + (long)a.H * HD, tid);
  }
}

// ------------------------------------------------------------------------------------------ dK, dV (+ dQ when fused)
// One block = NW waves x KW 16-key sub-tiles = NK keys of one (batch, head), swept over all queries in tiles of QT rows.
// Keys sit on the lanes (S = Q.K^T un-transposed), so P and dS are already the B operands of dV^T += dO^T.P and
// dK^T += Q^T.dS; dK and dV of the block's keys live in registers for the whole sweep (no atomics, deterministic).
//   BITS : the dropout keep-mask comes from the forward's bit words (AttnArgs::mbits) instead of the hash
//   FUSED: the block holds ALL keys of its (batch, head) (L_k <= NK), so dQ is complete inside the block too and the
//          separate dQ kernel (which recomputes S and dP: 7 GEMMs per tile instead of 5, Q/K/V/dO read twice) is not
//          launched.  dS crosses LDS once, transposed ([key][query] bf16, 8-byte stores); every wave then multiplies a
//          [16 query x HD/2] slice of dQ = dS.K over all NK keys (K^T fragments by transposed reads of the block's K tile)
//          and stores it.  delta = rowsum(dO * O) is computed here as well (8 lanes per query row).
// Per query row the Q image's 32 pad bytes carry the row's sideband: {lse', delta', keep-mask dwords [tile][half]}.
// PACKED: cu_seqlens launch (localize()); a separate instantiation because this kernel runs at the 256-register limit and the
// per-sample pointers of the packed view cost the padded launch 13 us of 82 (a2t backward at cfg 2) when they share one body
template <int HD, int NW, int KW, int QT, bool BITS, bool FUSED, bool PAIR = false, bool PACKED = false>
// two waves per SIMD (256 registers) wherever the kernel fits them without spilling inside its loops (`make check-isa` prints the
// spill counts per kernel and fails on scratch traffic inside a loop of a default-path kernel)
__global__ __launch_bounds__(NW * 64 * (PAIR ? 2 : 1), (!FUSED && !PAIR && HD <= 96 && NW == 4 && KW == 1) ? 3 : ((PAIR || (HD <= 96 && NW == 4 && (KW == 1 || (BITS && !FUSED) || FUSED))) ? 2 : 1)) void attn_bwd_dkv_kernel(const AttnArgs a0) {
  using G = AttnGeom<HD>;
  constexpr int KS = G::KS, DT = G::DT, STRIDE = G::STRIDE, NT = NW * 64;
  static_assert(QT == 32, "query tile");
  static_assert(!FUSED || (NW == 4 && DT % 2 == 0), "fused dQ: 4 waves, each a [16 x HD/2] slice");
  constexpr int DH = FUSED ? DT / 2 : 1;             // output tiles per wave of the fused dQ
  constexpr int NQS = QT / 16, NKQ = QT / 32;
  constexpr int NK = NW * KW * 16;                   // keys per block
  constexpr int NKTB = (NK + 63) / 64;               // 64-key tiles (mask words) a block touches
  constexpr int SB = G::HDP * 2;                     // sideband offset inside a Q-image row (its 32 pad bytes)
  static_assert(STRIDE - SB >= 8 + 8 * NKTB, "sideband must fit the pad bytes");
  constexpr int IMG = 2 * QT * STRIDE;               // Q image + dO image
  constexpr int DSS = QT * 2 + 32;                   // row stride of the dS^T tile ([key][QT queries] bf16)
  constexpr int KT_BYTES = FUSED ? NK * STRIDE : 0, DST_BYTES = FUSED ? NK * DSS : 0;
  constexpr int LDS_ONE = 2 * IMG + KT_BYTES + 2 * DST_BYTES;
  // PAIR: one workgroup of 2*NW waves = two independent (batch, head) problems, each with its own NW waves and its own LDS half;
  // they only share the barriers (same trip counts).  The fused kernel at 128 keys needs half the CU's LDS and a full register
  // budget per problem; two problems per workgroup keep two waves per SIMD with ONE workgroup per CU (exactly two 80 KB
  // workgroups would also fit, but a paired one does not depend on the dispatcher co-scheduling them).  (The wrong dS elements
  // once seen at two waves per SIMD were a packed subtract -- see the (dP - delta') line further down -- not the pairing.)
  __shared__ __attribute__((aligned(16))) char lds_all[LDS_ONE * (PAIR ? 2 : 1)];
  const int sub = PAIR ? (int)(threadIdx.x / (NW * 64)) : 0;
  char* const lds = lds_all + sub * LDS_ONE;
  char* const Ktile = lds + 2 * IMG;
  char* const dSt0 = Ktile + KT_BYTES;
  const int tid = PAIR ? (int)(threadIdx.x % (NW * 64)) : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
  int tile, bh;
  tile_and_head((a0.Lk + NK - 1) / NK, a0.B * a0.H, tile, bh, PAIR ? (int)blockIdx.x * 2 + sub : (int)blockIdx.x);
  const int b = bh / a0.H, h = bh - b * a0.H;
  const AttnArgs a = PACKED ? localize(a0, b, h) : a0;
  float* const cskv_row = a.cskv != nullptr ? a.cskv + (long)(b * gridDim_tiles(a0.Lk, NK) + tile) * (2L * a.H * HD) + h * HD : nullptr;
  if (!FUSED && tile * NK >= a.Lk) {                   // packed sequences: no key of this sample in the tile (fused: one tile holds them all)
    if (cskv_row != nullptr) for (int e = tid; e < HD; e += NT) { cskv_row[e] = 0.f; cskv_row[(long)a.H * HD + e] = 0.f; }
    return;
  }
  const int kbase = tile * NK + wave * KW * 16;

  bf16x8 kreg[FUSED ? 1 : KW][FUSED ? 1 : KS], vreg[KW][KS];
  bool kvalid[KW];
#pragma unroll
  for (int kw = 0; kw < KW; ++kw) {
    const int key = kbase + kw * 16 + i;
    const int kc = min(key, a.Lk - 1);
    kvalid[kw] = key < a.Lk && !(a.kpm != nullptr && a.kpm[(long)b * a.Lk + kc] != 0);
    const bf16_t* kp = a.K + ((long)b * a.Lk + kc) * a.ldk + h * HD;
    const bf16_t* vp = a.V + ((long)b * a.Lk + kc) * a.ldv + h * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int e = ks * 32 + 8 * g;
      if (!FUSED) kreg[kw][ks] = (e < HD) ? *(const bf16x8*)(kp + e) : zero8();    // fused: read from the block's K tile in LDS
      vreg[kw][ks] = (e < HD) ? *(const bf16x8*)(vp + e) : zero8();
    }
  }
  if (FUSED) load_tile<HD, NK, NT>(Ktile, a.K + (long)b * a.Lk * a.ldk + h * HD, a.ldk, 0, a.Lk, tid);    // rows >= L_k zero
  f32x4 dk[KW][DT], dv[KW][DT];
#pragma unroll
  for (int kw = 0; kw < KW; ++kw)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      dk[kw][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dv[kw][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  const float sl2 = a.scale * LOG2E;
  const uint32_t key32 = site_key(eff_seed(a.seed, a.seed_dev), a.site, (uint32_t)((a.b_offset + b) * a.H + h));
  uint32_t hkb[KW];                     // hash base of this lane's key: key32 + (key>>1)*CB  (a-term added per query)
#pragma unroll
  for (int kw = 0; kw < KW; ++kw) hkb[kw] = drop_base(key32, 0u, (uint32_t)((kbase + kw * 16 + i) >> 1));
  // BITS: this lane's key (sub-tile 0) sits at bit mbit0 of sideband dword mdw: tile = (kbase relative to the block's first key)
  // / 64, half = i >= 8; the same dword for every sub-tile of a wave when KW <= 2 (a wave's 32 keys never straddle a 64-key
  // tile).  Both are rebuilt from the lane id where they are used (a few VALU operations per tile) instead of living in two
  // registers across the loop: this kernel runs at the 256-register limit of two waves per SIMD.
  auto mask_pos = [&](int& mbit0, int& mdw) {
    int l2 = lane;
    asm volatile("" : "+v"(l2));        // opaque: keeps the rebuild inside the loop
    const int i2 = l2 & 15;
    mbit0 = ((i2 >> 2) * 16 + ((kbase & 63) >> 4) * 4 + (i2 & 3)) & 31;
    mdw = 2 + (((kbase - tile * NK) >> 6) * 2 + (i2 >> 3));
  };
  const int nkt_all = (a.Lk + 63) >> 6, kt0 = (tile * NK) >> 6;
  const bf16_t* Qb = a.Q + (long)b * a.Lq * a.ldq + h * HD;
  const bf16_t* dOb = a.dO + (long)b * a.Lq * a.lddo + h * HD;
  const bf16_t* Ob = a.O + (long)b * a.Lq * a.ldo + h * HD;
  const long lbase = ((long)b * a.H + h) * a.Lq;
  const int nqt = (a.Lq + QT - 1) / QT;

  const float l2ik = a.l2ik, keepfrac = a.keepfrac;
  TileRegs<HD, QT, NT> qr, dor;
  // sideband of query row `tid` of the tile in flight (threads tid < QT): lse', delta' (not fused), mask words
  float lse_r = 0.f, del_r = 0.f;
  unsigned long long mw_r[NKTB];
  // fused: delta = rowsum(dO * O), 8 threads per row (thread t: row t / 8, elements (t & 7) * HD/8 .. + HD/8).  The loads for
  // tile j+2 are issued at the top of iteration j and reduced right after its S / dP MFMAs (12 registers live only there); the
  // sum travels in one register to commit(j+2) at the top of iteration j+1.
  constexpr int EPT = HD / 8;
  float del_part = 0.f;
  auto delta_of = [&](int qt) -> float {
    const int q = min(qt * QT + (tid >> 3), a.Lq - 1);
    const bf16_t* dp_ = dOb + (long)q * a.lddo + (tid & 7) * EPT;
    const bf16_t* op_ = Ob + (long)q * a.ldo + (tid & 7) * EPT;
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < EPT / 2; ++e) {
      const bf16x2 x = *(const bf16x2*)(dp_ + 2 * e), y = *(const bf16x2*)(op_ + 2 * e);
      part += (float)x[0] * (float)y[0] + (float)x[1] * (float)y[1];
    }
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    part += __shfl_xor(part, 4);
    return part * keepfrac;
  };
  auto fetch = [&](int qt) {          // Q / dO rows of tile qt: global -> registers
    tile_fetch<HD, QT, NT>(qr, Qb, a.ldq, qt * QT, a.Lq, tid);
    tile_fetch<HD, QT, NT>(dor, dOb, a.lddo, qt * QT, a.Lq, tid);
  };
  auto fetch_side = [&](int qt) {     // sideband of tile qt: global -> registers (threads tid < QT; fused: delta by all threads)
    if (tid < QT) {
      const int q = qt * QT + tid;
      lse_r = q < a.Lq ? l2ik - a.lse[lbase + q] * LOG2E : -INFINITY;   // -inf -> p = 0 for rows past Lq
      if (!FUSED) del_r = q < a.Lq ? a.delta[lbase + q] * keepfrac : 0.f;
      if (BITS) {
#pragma unroll
        for (int t = 0; t < NKTB; ++t)
          mw_r[t] = (q < a.Lq && kt0 + t < nkt_all) ? a.mbits[(lbase + q) * nkt_all + kt0 + t] : 0ull;
      }
    }
    if (FUSED) del_part = delta_of(qt);
  };
  auto commit_side = [&](int buf) {   // sideband registers -> the pad bytes of image `buf` (which no wave is reading any more)
    char* base = lds + buf * IMG;
    if (tid < QT) {
      char* sb = base + tid * STRIDE + SB;
      *(LDS_PTR(float))(sb) = lse_r;
      if (!FUSED) *(LDS_PTR(float))(sb + 4) = del_r;
      if (BITS) {
#pragma unroll
        for (int t = 0; t < NKTB; ++t) {
          *(LDS_PTR(unsigned))(sb + 8 + 8 * t) = (unsigned)mw_r[t];
          *(LDS_PTR(unsigned))(sb + 12 + 8 * t) = (unsigned)(mw_r[t] >> 32);
        }
      }
    }
    if (FUSED && (tid & 7) == 0) {
      *(LDS_PTR(float))(base + (tid >> 3) * STRIDE + SB + 4) = del_part;        // (1 - p) * rowsum(dO * O); never leaves the block
    }
  };
  auto commit = [&](int buf) {        // Q / dO registers -> image `buf`
    char* base = lds + buf * IMG;
    tile_commit<HD, QT, NT>(qr, base, tid);
    tile_commit<HD, QT, NT>(dor, base + QT * STRIDE, tid);
  };
  // Pipeline of the query tiles: tile t is FETCHED (global -> registers) right after the barrier that closes iteration t-2,
  // COMMITTED (registers -> LDS image t & 1) right after the S / dP MFMAs of iteration t-1, and consumed in iteration t.  The
  // staging registers are therefore dead during the register-hungry part of an iteration (softmax, dS^T, dV / dK).
  fetch(0);
  fetch_side(0);
  commit(0);
  commit_side(0);
  if (nqt > 1) { fetch(1); fetch_side(1); commit_side(1); }
  __syncthreads();
  // fused: colsum(dQ) = (sum over queries of dS) . K -- one running sum per key instead of accumulators over the dQ slices
  float dsum[KW];
#pragma unroll
  for (int kw = 0; kw < KW; ++kw) dsum[kw] = 0.f;
  for (int qt = 0; qt < nqt; ++qt) {
    const char* Qt = lds + (qt & 1) * IMG;
    const char* dOt = Qt + QT * STRIDE;
    bf16x8 pf[KW][NKQ], dsf[KW][NKQ];
    char* const dSt = dSt0 + (qt & 1) * DST_BYTES;
#pragma unroll
    for (int qs = 0; qs < NQS; ++qs) {
      // S and dP of the 16 query rows qs*16 .. +15 against this wave's keys; the next sub-tile's MFMAs overlap this one's softmax
      f32x4 s[KW], dp[KW];
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        s[kw] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dp[kw] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 qfr = row_frag(Qt, STRIDE, qs * 16 + i, ks * 4 + g);
        const bf16x8 dofr = row_frag(dOt, STRIDE, qs * 16 + i, ks * 4 + g);
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          bf16x8 kf;
          if constexpr (FUSED) kf = row_frag(Ktile, STRIDE, wave * KW * 16 + kw * 16 + i, ks * 4 + g);
          else kf = kreg[kw][ks];
          s[kw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf, s[kw], 0, 0, 0);
          dp[kw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dofr, vreg[kw][ks], dp[kw], 0, 0, 0);
        }
      }
      if (qs == NQS - 1 && qt + 1 < nqt) commit((qt + 1) & 1);      // image last read in iteration qt-1, released by its barrier
      // sideband of query rows qs*16 + 4g + r: lse' = log2(1/(1-p_drop)) - lse*log2e, delta' = delta*(1-p_drop): pk = p/(1-p_drop)
      // straight from the exponent, P~ = keep ? pk : 0, dS = pk * (keep ? dP : 0  -  delta')
      float lse4[4], del4[4];
      unsigned mk4[4];
      int mbit0 = 0, mdw = 0;
      if (BITS) mask_pos(mbit0, mdw);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const char* sb = Qt + (qs * 16 + 4 * g + r) * STRIDE + SB;
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        const f32x2 ld2 = *(LDS_PTR(const f32x2))(sb);
        lse4[r] = ld2[0]; del4[r] = ld2[1];
        mk4[r] = BITS ? *(LDS_PTR(const unsigned))(sb + 4 * mdw) : 0u;
      }
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        const uint32_t key = (uint32_t)(kbase + kw * 16 + i);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pk = kvalid[kw] ? EXP2(fmaf(s[kw][r], sl2, lse4[r])) : 0.f;
          float pd = pk, dpd = dp[kw][r];
          if (BITS || a.thr16 != 0) {               // BITS kernels are only launched with dropout on
            bool keep;
            if (BITS) {
              keep = ((mk4[r] >> (mbit0 + 4 * kw)) & 1u) != 0u;     // the wave's sub-tiles are neighbours: n -> n + 1 is 4 bits up
            } else {
              // the lane owns ONE key (pair index key>>1, half key&1) and walks the queries: a-term by addition
              const uint32_t x = mix24(hkb[kw] + (uint32_t)(qt * QT + qs * 16 + 4 * g + r) * DROP_CA);
              keep = (key & 1u) ? keep_hi(x, a.thr16) : keep_lo(x, a.thr16);
            }
            pd = keep ? pk : 0.f;
            dpd = keep ? dpd : 0.f;
          }
          // Plain C on purpose.  This file is built with -fno-slp-vectorize: hipcc's SLP vectoriser turned the two key sub-tiles'
          // subtractions into  v_pk_add_f32 d, a, ld op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]  (ld = the (lse', delta') pair as
          // loaded), and on MI355X the LOW result of that form -- the half whose src1 operand is taken from the HIGH dword --
          // sporadically comes back as a.lo - 0 in lanes 48..63 (scripts_dev/forensics).  `make check-isa` fails
          // the build if the form shows up in any code object again.
          const float dif = dpd - del4[r];
          const float dsv = pk * dif;
          if (FUSED) dsum[kw] += dsv;
          pf[kw][qs >> 1][(qs & 1) * 4 + r] = (bf16_t)pd;
          dsf[kw][qs >> 1][(qs & 1) * 4 + r] = (bf16_t)dsv;
        }
        // dS^T[key][queries qs*16 + 4g .. +3]: one 8-byte store, straight from the half of the operand register just filled
        if (FUSED) {
          const bf16x8 f = dsf[kw][qs >> 1];
          const bf16x4 h4 = (qs & 1) ? (bf16x4){f[4], f[5], f[6], f[7]} : (bf16x4){f[0], f[1], f[2], f[3]};
          *(LDS_PTR(bf16x4))(dSt + (wave * KW * 16 + kw * 16 + i) * DSS + (qs * 16 + 4 * g) * 2) = h4;
        }
      }
    }
#pragma unroll
    for (int kq = 0; kq < NKQ; ++kq)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bf16x8 dotf = tr_frag(dOt, STRIDE, 32 * kq, dt * 16, lane);
        const bf16x8 qtf = tr_frag(Qt, STRIDE, 32 * kq, dt * 16, lane);
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          dv[kw][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dotf, pf[kw][kq], dv[kw][dt], 0, 0, 0);
          dk[kw][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf[kw][kq], dk[kw][dt], 0, 0, 0);
        }
        if (FUSED && KW == 2 && (dt & 1)) __builtin_amdgcn_sched_barrier(0);      // at most two tiles' fragments in flight (registers)
      }
    __syncthreads();          // every read of this tile's Q / dO image is done; (fused) every wave's dS^T is in LDS
    // image qt & 1 is released: start tile qt+2 (its Q / dO rows stay in registers until the S / dP MFMAs of the next iteration
    // are issued; its sideband goes to the pad bytes of the released image behind the dQ phase, off the register-hungry part)
    if (qt + 2 < nqt) { fetch(qt + 2); fetch_side(qt + 2); }
    if constexpr (FUSED) {
      // dQ[16 queries x HD/2] of this wave: query sub-tile wave>>1, output tiles (wave&1)*DT/2 .. ; contraction over all NK keys in
      // the permuted k-order of tr_frag on BOTH operands.  Computed transposed (rows = head dim) so a lane owns 4 consecutive
      // columns of one query row: 8-byte stores.  The dS^T buffer alternates per tile, so the next tile's stores cannot reach it.
      const int qsub = wave >> 1, dt0 = (wave & 1) * DH;
      f32x4 dq[DH];
#pragma unroll
      for (int t = 0; t < DH; ++t) dq[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < NK / 32; ++kk) {
        const bf16x8 dsT = tr_frag(dSt, DSS, 32 * kk, qsub * 16, lane);
#pragma unroll
        for (int t = 0; t < DH; ++t) {
          const bf16x8 kT = tr_frag(Ktile, STRIDE, 32 * kk, (dt0 + t) * 16, lane);
          dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kT, dsT, dq[t], 0, 0, 0);
        }
      }
      const int q = qt * QT + qsub * 16 + i;
      if (q < a.Lq) {
        bf16_t* dqp = a.dQ + ((long)b * a.Lq + q) * a.lddq + h * HD + 4 * g;
#pragma unroll
        for (int t = 0; t < DH; ++t) {
          bf16x4 w;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            w[r] = (bf16_t)(dq[t][r] * a.scale);
          }
          *(bf16x4*)(dqp + (dt0 + t) * 16) = w;
        }
      }
    }
    if (qt + 2 < nqt) commit_side(qt & 1);
  }
  f32x4 csk[DT], csv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { csk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; csv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  int tid_e = PAIR ? (int)(threadIdx.x % (NW * 64)) : (int)threadIdx.x;
  asm volatile("" : "+v"(tid_e));       // opaque: the epilogue's lane-derived addresses are rebuilt here, not carried through the loop
  const int i_e = tid_e & 15, g_e = (tid_e >> 4) & 3, kbase_e = tile * NK + (tid_e >> 6) * KW * 16;
#pragma unroll
  for (int kw = 0; kw < KW; ++kw) {
    const int key = kbase_e + kw * 16 + i_e;
    if (key >= a.Lk) continue;
    bf16_t* dkp = a.dK + ((long)b * a.Lk + key) * a.lddk + h * HD + 4 * g_e;
    bf16_t* dvp = a.dV + ((long)b * a.Lk + key) * a.lddv + h * HD + 4 * g_e;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      bf16x4 wk, wv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float vk = dk[kw][dt][r] * a.scale, vv = dv[kw][dt][r];
        wk[r] = (bf16_t)vk;
        wv[r] = (bf16_t)vv;
        csk[dt][r] += vk;
        csv[dt][r] += vv;
      }
      *(bf16x4*)(dkp + dt * 16) = wk;
      *(bf16x4*)(dvp + dt * 16) = wv;
    }
  }
  if (a.cskv != nullptr) {         // kernel-uniform
    float* row = cskv_row;
    __syncthreads();
    block_colsum_store<DT, NW>(csk, (float*)lds, row, tid);
    block_colsum_store<DT, NW>(csv, (float*)lds + NW * DT * 16, row + (long)a.H * HD, tid);
  }
  if constexpr (FUSED) {
    if (a.csq != nullptr) {  // column sums of the block's dQ (one partial row per (batch, head)): scale * sum_key dsum[key] * K[key][:]
      float* red = (float*)lds + 2 * NW * DT * 16;
      float c[KS][8];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[ks][j] = 0.f;
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        float t = dsum[kw];                       // this lane's queries only: add the other three lane groups of the key
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          {
            const bf16x8 kf = row_frag(Ktile, STRIDE, wave * KW * 16 + kw * 16 + i, ks * 4 + g);
#pragma unroll
            for (int j = 0; j < 8; ++j) c[ks][j] += t * (float)kf[j];
          }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = c[ks][j];
#pragma unroll
          for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);      // over the 16 keys of the lane group
          c[ks][j] = v;
        }
      __syncthreads();
      if (i == 0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int j = 0; j < 8; ++j) red[wave * HD + ks * 32 + 8 * g + j] = c[ks][j];
      }
      __syncthreads();
      for (int e = tid; e < HD; e += NW * 64) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w * HD + e];
        a.csq[(long)b * ((long)a.H * HD) + h * HD + e] = t * a.scale;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ export
// Head-averaged (post-dropout when thr16 != 0) probabilities [B, Lq, Lk] from the saved LSE
// (reference: need_weights=True path of nn.MultiheadAttention, average_attn_weights=True).
// Inference/analysis only: plain VALU dot products, 8 query rows x 64 keys per block.
template <int HD>
__global__ __launch_bounds__(64) void attn_probs_kernel(const AttnArgs a) {
  __shared__ float qs[8][HD];
  const int tid = threadIdx.x;
  const int b = blockIdx.z, q0 = blockIdx.y * 8, key = blockIdx.x * 64 + tid;
  const bool kin = key < a.Lk;
  const bool pad = !kin || (a.kpm != nullptr && a.kpm[(long)b * a.Lk + key] != 0);
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = 0.f;
  for (int h = 0; h < a.H; ++h) {
    __syncthreads();
    for (int id = tid; id < 8 * HD; id += 64) {
      const int r = id / HD, e = id - r * HD;
      const int q = min(q0 + r, a.Lq - 1);
      qs[r][e] = (float)a.Q[((long)b * a.Lq + q) * a.ldq + h * HD + e];
    }
    __syncthreads();
    float dot[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) dot[r] = 0.f;
    if (kin) {
      const bf16_t* kp = a.K + ((long)b * a.Lk + key) * a.ldk + h * HD;
      for (int c = 0; c < HD / 8; ++c) {
        const bf16x8 kv = *(const bf16x8*)(kp + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float kf = (float)kv[j];
#pragma unroll
          for (int r = 0; r < 8; ++r) dot[r] += kf * qs[r][c * 8 + j];
        }
      }
    }
    const uint32_t key32 = site_key(eff_seed(a.seed, a.seed_dev), a.site, (uint32_t)((a.b_offset + b) * a.H + h));
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int q = min(q0 + r, a.Lq - 1);
      // a query whose keys are ALL PAD has lse = -inf: the reference's softmax over an all -inf row is NaN in every
      // column (and this kernel family's O is NaN there too), so the exported map says NaN as well, not 0
      const float lse_q = a.lse[((long)b * a.H + h) * a.Lq + q];
      float p = (lse_q == -INFINITY) ? __builtin_nanf("") : (pad ? 0.f : __expf(dot[r] * a.scale - lse_q));
      if (a.thr16 != 0) p = keep16(key32, (uint32_t)q, (uint32_t)key, a.thr16) ? p * a.inv_keep : 0.f;
      acc[r] += p;
    }
  }
  if (kin) {
    const float invH = 1.f / (float)a.H;
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (q0 + r < a.Lq) a.probs[((long)b * a.Lq + q0 + r) * a.Lk + key] = acc[r] * invH;
  }
}

// ------------------------------------------------------------------------------------------ host
// Sub-tiles per wave.  Measured on cfg 2 (profiles/): the forward is fastest with two 16-row query sub-tiles
// per wave (K/V fragment reuse), the backward kernels with one (128 instead of ~220 VGPRs -> twice the
// waves per SIMD to cover their long VALU chains, and less padding waste at L=400).
static int attn_wide(int backward) { return backward ? 0 : 1; }      // (the backward's width is chosen per launch: bwd_wide)
static int check_common(const AttnArgs& a, int hd) {
  HRIEMO_CHECK(a.B > 0 && a.H > 0 && a.Lq > 0 && a.Lk > 0, "attn: empty problem");
  HRIEMO_CHECK(hd == 16 || hd == 32 || hd == 64 || hd == 96 || hd == 128, "attn: head_dim %d not built (16/32/64/96/128)", hd);
  HRIEMO_CHECK(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0, "attn: leading dims must be multiples of 8");
  HRIEMO_CHECK(((uintptr_t)a.Q % 16) == 0 && ((uintptr_t)a.K % 16) == 0 && ((uintptr_t)a.V % 16) == 0, "attn: unaligned Q/K/V");
  return 0;
}

#define DISPATCH_HD(hd, CALL)               \
  switch (hd) {                             \
    case 16: { CALL(16); } break;           \
    case 32: { CALL(32); } break;           \
    case 64: { CALL(64); } break;           \
    case 96: { CALL(96); } break;           \
    case 128: { CALL(128); } break;         \
  }

static void fill_drop(AttnArgs& a, float p, uint64_t seed, const unsigned long long* seed_dev, uint32_t site, int b_offset) {
  DropCfg d = make_drop(p, seed, site);
  a.thr16 = d.thr16; a.inv_keep = d.inv_keep; a.seed = seed; a.site = site; a.b_offset = b_offset; a.seed_dev = seed_dev;
  a.l2ik = d.thr16 != 0 ? log2f(d.inv_keep) : 0.f;
  a.keepfrac = d.thr16 != 0 ? 1.f / d.inv_keep : 1.f;
}

extern "C" long hriemo_attn_mask_bytes(int B, int H, int Lq, int Lk) { return (long)B * H * Lq * ((Lk + 63) / 64) * 8; }

static int check_packed(const AttnArgs& a) {
  HRIEMO_CHECK((a.cu_q == nullptr) == (a.cu_k == nullptr), "attn: cu_seqlens_q and cu_seqlens_k must be given together");
  HRIEMO_CHECK(a.cu_q == nullptr || a.kpm == nullptr, "attn: packed sequences carry their lengths, a key_padding_mask cannot be combined with them");
  return 0;
}

static int attn_fwd_impl(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, void* O,
                         long ldo, const unsigned char* key_padding_mask, float* lse, int B, int H, int Lq,
                         int Lk, int head_dim, float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                         unsigned site, int b_offset, void* drop_mask_bits, const int* cu_q, const int* cu_k, hipStream_t st,
                         void* Oq = nullptr, long ldoq = 0, void* So = nullptr, long ldso = 0) {
  AttnArgs a = {};
  a.cu_q = cu_q; a.cu_k = cu_k;
  a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.O = (bf16_t*)O; a.ldo = ldo;
  a.Oq = (uint8_t*)Oq; a.ldoq = ldoq; a.So = (uint8_t*)So; a.ldso = ldso;
  HRIEMO_CHECK(Oq == nullptr || (cu_q == nullptr && head_dim % 32 == 0 && So != nullptr && ldoq % 4 == 0 && ((uintptr_t)Oq % 4) == 0 &&
                                 ldso >= (long)B * Lq),
               "attn_fwd: the MX-fp8 copy of O needs padded rows, head_dim %% 32 == 0 and a scale buffer of >= B*Lq columns");
  a.kpm = key_padding_mask; a.lse = lse; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk;
  a.scale = 1.0f / sqrtf((float)head_dim);
  fill_drop(a, p_drop, seed, seed_dev, site, b_offset);
  a.mbits = (unsigned long long*)drop_mask_bits;
  if (check_common(a, head_dim) || check_packed(a)) return 1;
  HRIEMO_CHECK(ldo % 4 == 0 && ((uintptr_t)O % 8) == 0 && ((uintptr_t)drop_mask_bits % 8) == 0, "attn_fwd: unaligned O / mask bits");
  hriemo_prof_begin(HP_ATTN_FWD, st);
  // two 16-row query sub-tiles per wave (K/V fragment reuse) unless the key loop is short and the 128-row tiles pad the
  // query side visibly more than 64-row tiles do (L_q = 400, L_k = 128: 512 vs 448 rows, 44.3 vs 40.5 us)
  const bool short_keys_padded = Lk <= 128 && ((Lq + 63) / 64) * 64 * 20 < ((Lq + 127) / 128) * 128 * 19;
  const bool wb = a.mbits != nullptr && a.thr16 != 0;
#define FWD(HD, NW_, QW_, GRID, THREADS)                                                                         \
  if (wb) hipLaunchKernelGGL((attn_fwd_kernel<HD, NW_, QW_, true>), dim3(GRID), dim3(THREADS), 0, st, a);         \
  else hipLaunchKernelGGL((attn_fwd_kernel<HD, NW_, QW_, false>), dim3(GRID), dim3(THREADS), 0, st, a)
  if (Lq > 64 && attn_wide(0) && !short_keys_padded) {
#define CALL(HD) FWD(HD, 4, 2, ((Lq + 127) / 128) * B * H, 256)
    DISPATCH_HD(head_dim, CALL)
#undef CALL
  } else if (Lq > 16) {
#define CALL(HD) FWD(HD, 4, 1, ((Lq + 63) / 64) * B * H, 256)
    DISPATCH_HD(head_dim, CALL)
#undef CALL
  } else {
#define CALL(HD) FWD(HD, 1, 1, B * H, 64)
    DISPATCH_HD(head_dim, CALL)
#undef CALL
  }
#undef FWD
  HRIEMO_LAUNCH_CHECK("attn_fwd_kernel");
  hriemo_prof_end(HP_ATTN_FWD, st, 4.0 * B * H * (double)Lq * Lk * head_dim);
  return 0;
}

extern "C" int hriemo_attn_fwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, void* O,
                               long ldo, const unsigned char* key_padding_mask, float* lse, int B, int H, int Lq,
                               int Lk, int head_dim, float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                               unsigned site, int b_offset, void* drop_mask_bits, hipStream_t st) {
  return attn_fwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, key_padding_mask, lse, B, H, Lq, Lk, head_dim, p_drop, seed, seed_dev, site,
                       b_offset, drop_mask_bits, nullptr, nullptr, st);
}
// hriemo_attn_fwd that also leaves the MX-fp8 form of O (bytes Oq[B*Lq][ldoq], E8M0 scales So[H*hd/32][ldso]) for the
// out-projection GEMM of the fp8 mode: no separate quantisation pass over O
extern "C" int hriemo_attn_fwd_q(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, void* O,
                                 long ldo, const unsigned char* key_padding_mask, float* lse, int B, int H, int Lq,
                                 int Lk, int head_dim, float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                                 unsigned site, int b_offset, void* drop_mask_bits, void* Oq, long ldoq, void* So, long ldso,
                                 hipStream_t st) {
  HRIEMO_CHECK(Oq != nullptr, "attn_fwd_q: Oq required");
  return attn_fwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, key_padding_mask, lse, B, H, Lq, Lk, head_dim, p_drop, seed, seed_dev, site,
                       b_offset, drop_mask_bits, nullptr, nullptr, st, Oq, ldoq, So, ldso);
}
extern "C" int hriemo_attn_fwd_varlen(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, void* O,
                                      long ldo, const int* cu_seqlens_q, const int* cu_seqlens_k, float* lse, int B, int H,
                                      int max_len_q, int max_len_k, int head_dim, float p_drop, unsigned long long seed,
                                      const unsigned long long* seed_dev, unsigned site, int b_offset, void* drop_mask_bits,
                                      hipStream_t st) {
  HRIEMO_CHECK(cu_seqlens_q != nullptr && cu_seqlens_k != nullptr, "attn_fwd_varlen: cu_seqlens missing");
  return attn_fwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, nullptr, lse, B, H, max_len_q, max_len_k, head_dim, p_drop, seed, seed_dev,
                       site, b_offset, drop_mask_bits, cu_seqlens_q, cu_seqlens_k, st);
}

// Backward tile width for a row side of length L (queries for dQ, keys for dK/dV).  Narrow blocks (64 rows, ~160
// VGPRs, 3 per CU) win in general; but their blocks live as long as the whole key / query loop, so a grid that fills
// the chip 1.33 times (B*H = 512, L = 128: 1024 blocks on 768 slots) runs a second, mostly empty round.  The wide tile
// (128 rows, 2 blocks per CU) is picked when it fills its slots >= 90 % and the narrow one < 75 %
// (scripts_dev/bench_attn_split.py: dK/dV at L_k = 128 51.0 vs 60.5 us, dQ at L_q = 128 39.8 vs 43.5 us).
static int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}
static bool bwd_wide(int L, int BH, int head_dim) {
  if (L <= 64) return false;
  if (attn_wide(1)) return true;
  if (head_dim > 96) return false;                       // register budgets above were measured for hd <= 96
  const long sn = 3L * cu_count(), sw = 2L * cu_count();
  const long nn = (long)((L + 63) / 64) * BH, nw = (long)((L + 127) / 128) * BH;
  const double en = (double)nn / (double)(((nn + sn - 1) / sn) * sn), ew = (double)nw / (double)(((nw + sw - 1) / sw) * sw);
  return en < 0.75 && ew >= 0.9;
}
// Single-pass backward: one block holds all keys of a (batch, head) (16 < L_k <= 128) and produces dQ, dK and dV together
// (5 GEMMs per tile instead of 7, Q/K/V/dO read once): a2t backward 82 us instead of 114 at cfg 2.
// Its first build returned a few wrong dS elements per
// launch with the bit-word mask at two waves per SIMD; traced to the compiler's packed form of the (dP - delta') subtraction
// (v_pk_add_f32 with the low result reading the high dword of src1: round 3's scripts_dev/forensics reproduce it in 20 of 20
// runs and isolate the operand select); the file is compiled without SLP vectorisation since.
static bool bwd_fused(int Lk, int head_dim, int B, int H) {
  // (one 256-thread workgroup per (batch, head).  Round 2 first shipped heads 2j, 2j+1 of a sample in ONE 512-thread workgroup:
  // its block-wide barriers coupled the two problems -- a2t backward 80 us against 66 -- and that instantiation spilled inside
  // its loops; removed in round 4)
  (void)B; (void)H;
  return Lk > 16 && Lk <= 128 && head_dim >= 32;
}

extern "C" int hriemo_attn_bwd_colsum_rows(int B, int H, int L, int head_dim);
extern "C" int hriemo_attn_bwd_single_pass(int B, int H, int Lk, int head_dim) { return bwd_fused(Lk, head_dim, B, H) ? 1 : 0; }

// Query-resident single pass (attn_bwd_qres_kernel): all queries of a (batch, head) in one block, the keys swept -- for
// 16 < L_q <= 128 < L_k (t2a at cfg 2; the key-resident form above takes L_k <= 128).
static bool bwd_qres(int Lq, int Lk, int head_dim, int B, int H) {
  return !bwd_fused(Lk, head_dim, B, H) && Lq > 16 && Lq <= 128 && head_dim >= 32;
}
// 1 if the backward of this shape is ONE kernel of either form (the forward then writes the dropout keep-mask as bit words)
extern "C" int hriemo_attn_bwd_single_pass_q(int B, int H, int Lq, int Lk, int head_dim) {
  return (bwd_fused(Lk, head_dim, B, H) || bwd_qres(Lq, Lk, head_dim, B, H)) ? 1 : 0;
}
// rows of the dK | dV column-sum partials when both lengths are known (the query-resident kernel leaves one row per batch)
extern "C" int hriemo_attn_bwd_kv_colsum_rows(int B, int H, int Lq, int Lk, int head_dim) {
  if (bwd_qres(Lq, Lk, head_dim, B, H)) return B;
  return hriemo_attn_bwd_colsum_rows(B, H, Lk, head_dim);
}

// rows of the column-sum partials hriemo_attn_bwd leaves behind: dK|dV side (sequence of length Lk) ...
extern "C" int hriemo_attn_bwd_colsum_rows(int B, int H, int L, int head_dim) {
  if (bwd_fused(L, head_dim, B, H)) return B;
  if (bwd_wide(L, B * H, head_dim)) return B * ((L + 127) / 128);
  if (L > 16) return B * ((L + 63) / 64);
  return B;
}
// ... and dQ side (depends on both lengths: the fused kernel writes one row per (batch, head))
extern "C" int hriemo_attn_bwd_dq_colsum_rows(int B, int H, int Lq, int Lk, int head_dim) {
  if (bwd_fused(Lk, head_dim, B, H) || bwd_qres(Lq, Lk, head_dim, B, H)) return B;
  if (bwd_wide(Lq, B * H, head_dim)) return B * ((Lq + 127) / 128);
  if (Lq > 16) return B * ((Lq + 63) / 64);
  return B;
}

#define DISPATCH_HD_EVEN(hd, CALL)          \
  switch (hd) {                             \
    case 32: { CALL(32); } break;           \
    case 64: { CALL(64); } break;           \
    case 96: { CALL(96); } break;           \
    case 128: { CALL(128); } break;         \
  }

static int attn_bwd_impl(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv,
                         const void* O, long ldo, const void* dO, long lddo, void* dQ, long lddq, void* dK,
                         long lddk, void* dV, long lddv, const unsigned char* key_padding_mask,
                         const float* lse, float* delta, int B, int H, int Lq, int Lk, int head_dim,
                         float p_drop, unsigned long long seed, const unsigned long long* seed_dev, unsigned site, int b_offset,
                         float* dq_colsum_partials, float* dkv_colsum_partials, const void* drop_mask_bits, const int* cu_q,
                         const int* cu_k, hipStream_t st) {
  AttnArgs a = {};
  a.cu_q = cu_q; a.cu_k = cu_k;
  a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.O = (bf16_t*)O; a.ldo = ldo;
  a.csq = dq_colsum_partials; a.cskv = dkv_colsum_partials;
  a.dO = (const bf16_t*)dO; a.lddo = lddo;
  a.dQ = (bf16_t*)dQ; a.dK = (bf16_t*)dK; a.dV = (bf16_t*)dV; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.kpm = key_padding_mask; a.lse = (float*)lse; a.delta = delta; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk;
  a.scale = 1.0f / sqrtf((float)head_dim);
  fill_drop(a, p_drop, seed, seed_dev, site, b_offset);
  a.mbits = (unsigned long long*)drop_mask_bits;
  if (check_common(a, head_dim) || check_packed(a)) return 1;
  HRIEMO_CHECK(ldo % 8 == 0 && lddo % 8 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0, "attn_bwd: bad leading dims");
  HRIEMO_CHECK(((uintptr_t)O % 16) == 0 && ((uintptr_t)dO % 16) == 0 && ((uintptr_t)dQ % 8) == 0 &&
                   ((uintptr_t)dK % 8) == 0 && ((uintptr_t)dV % 8) == 0 && ((uintptr_t)drop_mask_bits % 8) == 0, "attn_bwd: unaligned operand");
  const bool bits = a.thr16 != 0 && a.mbits != nullptr;
  if (bwd_fused(Lk, head_dim, B, H)) {
    hriemo_prof_begin(HP_ATTN_BWD_DKV, st);
    // one 256-thread workgroup per (batch, head); two of them share a CU where two problems fit its LDS
#define CALLF(HD, KW_, BITS_)                                                                                                    \
  {                                                                                                                              \
    if (a.cu_q != nullptr) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 4, KW_, 32, BITS_, true, false, true>), dim3(B * H), dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 4, KW_, 32, BITS_, true, false>), dim3(B * H), dim3(256), 0, st, a);          \
  }
    if (Lk <= 64) {
      if (bits) {
#define CALL(HD) CALLF(HD, 1, true)
        DISPATCH_HD_EVEN(head_dim, CALL)
#undef CALL
      } else {
#define CALL(HD) CALLF(HD, 1, false)
        DISPATCH_HD_EVEN(head_dim, CALL)
#undef CALL
      }
    } else {
      if (bits) {
#define CALL(HD) CALLF(HD, 2, true)
        DISPATCH_HD_EVEN(head_dim, CALL)
#undef CALL
      } else {
#define CALL(HD) CALLF(HD, 2, false)
        DISPATCH_HD_EVEN(head_dim, CALL)
#undef CALL
      }
    }
#undef CALLF
    HRIEMO_LAUNCH_CHECK("attn_bwd_dkv_kernel (fused dQ)");
    hriemo_prof_end(HP_ATTN_BWD_DKV, st, 10.0 * B * H * (double)Lq * Lk * head_dim);
    return 0;
  }
  if (bwd_qres(Lq, Lk, head_dim, B, H)) {
    hriemo_prof_begin(HP_ATTN_BWD_DKV, st);
    if (bits) {
#define CALL(HD) hipLaunchKernelGGL((attn_bwd_qres_kernel<HD, true>), dim3(B * H), dim3(512), 0, st, a)
      DISPATCH_HD_EVEN(head_dim, CALL)
#undef CALL
    } else {
#define CALL(HD) hipLaunchKernelGGL((attn_bwd_qres_kernel<HD, false>), dim3(B * H), dim3(512), 0, st, a)
      DISPATCH_HD_EVEN(head_dim, CALL)
#undef CALL
    }
    HRIEMO_LAUNCH_CHECK("attn_bwd_qres_kernel");
    hriemo_prof_end(HP_ATTN_BWD_DKV, st, 10.0 * B * H * (double)Lq * Lk * head_dim);
    return 0;
  }
  hriemo_prof_begin(HP_ATTN_BWD_DQ, st);
  if (bwd_wide(Lq, B * H, head_dim)) {
    if (bits) {
#define CALL(HD) hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, 4, 2, true>), dim3(((Lq + 127) / 128) * B * H), dim3(256), 0, st, a)
      DISPATCH_HD(head_dim, CALL)
#undef CALL
    } else {
#define CALL(HD) hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, 4, 2, false>), dim3(((Lq + 127) / 128) * B * H), dim3(256), 0, st, a)
      DISPATCH_HD(head_dim, CALL)
#undef CALL
    }
  } else if (Lq > 16) {
    if (bits) {
#define CALL(HD) hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, 4, 1, true>), dim3(((Lq + 63) / 64) * B * H), dim3(256), 0, st, a)
      DISPATCH_HD(head_dim, CALL)
#undef CALL
    } else {
#define CALL(HD) hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, 4, 1, false>), dim3(((Lq + 63) / 64) * B * H), dim3(256), 0, st, a)
      DISPATCH_HD(head_dim, CALL)
#undef CALL
    }
  } else {
    if (bits) {
#define CALL(HD) hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, 1, 1, true>), dim3(B * H), dim3(64), 0, st, a)
      DISPATCH_HD(head_dim, CALL)
#undef CALL
    } else {
#define CALL(HD) hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, 1, 1, false>), dim3(B * H), dim3(64), 0, st, a)
      DISPATCH_HD(head_dim, CALL)
#undef CALL
    }
  }
  HRIEMO_LAUNCH_CHECK("attn_bwd_dq_kernel");
  hriemo_prof_end(HP_ATTN_BWD_DQ, st, 6.0 * B * H * (double)Lq * Lk * head_dim);
  hriemo_prof_begin(HP_ATTN_BWD_DKV, st);
  if (bwd_wide(Lk, B * H, head_dim)) {
    if (bits) {
#define CALL(HD)                                                                                                             \
  if (a.cu_q != nullptr) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 4, 2, 32, true, false, false, true>), dim3(((Lk + 127) / 128) * B * H), dim3(256), 0, st, a); \
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 4, 2, 32, true, false>), dim3(((Lk + 127) / 128) * B * H), dim3(256), 0, st, a)
      DISPATCH_HD(head_dim, CALL)
#undef CALL
    } else {
#define CALL(HD) hipLaunchKernelGGL((attn_bwd_dkv_hash_kernel<HD, 4, 2, 32>), dim3(((Lk + 127) / 128) * B * H), dim3(256), 0, st, a)
      DISPATCH_HD(head_dim, CALL)
#undef CALL
    }
  } else if (Lk > 16) {
    if (bits) {
#define CALL(HD)                                                                                                             \
  if (a.cu_q != nullptr) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 4, 1, 32, true, false, false, true>), dim3(((Lk + 63) / 64) * B * H), dim3(256), 0, st, a); \
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 4, 1, 32, true, false>), dim3(((Lk + 63) / 64) * B * H), dim3(256), 0, st, a)
      DISPATCH_HD(head_dim, CALL)
#undef CALL
    } else {
#define CALL(HD) hipLaunchKernelGGL((attn_bwd_dkv_hash_kernel<HD, 4, 1, 32>), dim3(((Lk + 63) / 64) * B * H), dim3(256), 0, st, a)
      DISPATCH_HD(head_dim, CALL)
#undef CALL
    }
  } else {
    if (bits) {
#define CALL(HD)                                                                                                             \
  if (a.cu_q != nullptr) hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 1, 1, 32, true, false, false, true>), dim3(B * H), dim3(64), 0, st, a); \
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, 1, 1, 32, true, false>), dim3(B * H), dim3(64), 0, st, a)
      DISPATCH_HD(head_dim, CALL)
#undef CALL
    } else {
#define CALL(HD) hipLaunchKernelGGL((attn_bwd_dkv_hash_kernel<HD, 1, 1, 32>), dim3(B * H), dim3(64), 0, st, a)
      DISPATCH_HD(head_dim, CALL)
#undef CALL
    }
  }
  HRIEMO_LAUNCH_CHECK("attn_bwd_dkv_kernel");
  hriemo_prof_end(HP_ATTN_BWD_DKV, st, 8.0 * B * H * (double)Lq * Lk * head_dim);
  return 0;
}

extern "C" int hriemo_attn_bwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv,
                               const void* O, long ldo, const void* dO, long lddo, void* dQ, long lddq, void* dK,
                               long lddk, void* dV, long lddv, const unsigned char* key_padding_mask,
                               const float* lse, float* delta, int B, int H, int Lq, int Lk, int head_dim,
                               float p_drop, unsigned long long seed, const unsigned long long* seed_dev, unsigned site, int b_offset,
                               float* dq_colsum_partials, float* dkv_colsum_partials, const void* drop_mask_bits, hipStream_t st) {
  return attn_bwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, dQ, lddq, dK, lddk, dV, lddv, key_padding_mask, lse, delta, B, H, Lq,
                       Lk, head_dim, p_drop, seed, seed_dev, site, b_offset, dq_colsum_partials, dkv_colsum_partials, drop_mask_bits,
                       nullptr, nullptr, st);
}
extern "C" int hriemo_attn_bwd_varlen(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv,
                                      const void* O, long ldo, const void* dO, long lddo, void* dQ, long lddq, void* dK,
                                      long lddk, void* dV, long lddv, const int* cu_seqlens_q, const int* cu_seqlens_k,
                                      const float* lse, float* delta, int B, int H, int max_len_q, int max_len_k, int head_dim,
                                      float p_drop, unsigned long long seed, const unsigned long long* seed_dev, unsigned site,
                                      int b_offset, float* dq_colsum_partials, float* dkv_colsum_partials,
                                      const void* drop_mask_bits, hipStream_t st) {
  HRIEMO_CHECK(cu_seqlens_q != nullptr && cu_seqlens_k != nullptr, "attn_bwd_varlen: cu_seqlens missing");
  return attn_bwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, dQ, lddq, dK, lddk, dV, lddv, nullptr, lse, delta, B, H, max_len_q,
                       max_len_k, head_dim, p_drop, seed, seed_dev, site, b_offset, dq_colsum_partials, dkv_colsum_partials,
                       drop_mask_bits, cu_seqlens_q, cu_seqlens_k, st);
}

extern "C" int hriemo_attn_probs(const void* Q, long ldq, const void* K, long ldk, const unsigned char* key_padding_mask,
                                 const float* lse, float* probs, int B, int H, int Lq, int Lk, int head_dim,
                                 float p_drop, unsigned long long seed, const unsigned long long* seed_dev, unsigned site, int b_offset,
                               hipStream_t st) {
  AttnArgs a = {};
  a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)K;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldk;
  a.kpm = key_padding_mask; a.lse = (float*)lse; a.probs = probs; a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk;
  a.scale = 1.0f / sqrtf((float)head_dim);
  fill_drop(a, p_drop, seed, seed_dev, site, b_offset);
  if (check_common(a, head_dim)) return 1;
#define CALL(HD) hipLaunchKernelGGL((attn_probs_kernel<HD>), dim3((Lk + 63) / 64, (Lq + 7) / 8, B), dim3(64), 0, st, a)
  DISPATCH_HD(head_dim, CALL)
#undef CALL
  HRIEMO_LAUNCH_CHECK("attn_probs_kernel");
  return 0;
}
