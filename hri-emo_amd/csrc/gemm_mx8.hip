// MX-fp8 GEMM for gfx950 (BASELINE configs[4]: "fp8 MFMA path"): C[M,N] = A[M,K] . B[N,K]^T with both operands stored as
// OCP e4m3 bytes + one E8M0 scale per 32 k-elements (OCP microscaling, block size 32), multiplied on
// v_mfma_scale_f32_16x16x128_f8f6f4 -- the only fp8 form that runs above the bf16 rate on this chip (2x per clock; the
// plain fp8 16x16x32 runs AT the bf16 rate: scripts_dev/mfma_rate.hip) -- fp32 accumulate, the same fused epilogues
// (bias, ReLU, ReLU mask, residual add) and the same persistent, LDS-DMA-fed structure as the bf16 kernel (gemm.hip).
// Sites: every nn.Linear / packed in-projection / out-projection of the fusion blocks and the decoder in the FORWARD
// direction (models/cross_modal_block_tacfn.py:24-52, models/emotion_decoder.py:14-27); the backward GEMMs stay bf16.
//
// Layout notes (measured on hardware by scripts_dev/mx8_probe*.hip, pinned by tests/test_gpu_mx8.py):
//  * operand lane map of the 128-deep step: lane l = (i = l & 15, g = l >> 4) holds row (column) i and, in registers 0-3,
//    k = 16g .. 16g+15, in registers 4-7, k = 64+16g .. 64+16g+15 (the instruction is two 64-deep halves side by side);
//    the scale byte of lane (i, g) -- the byte of its scale VGPR that opsel names -- applies to row i, k = 32g .. 32g+31,
//    i.e. to bytes held by lanes (i, 2(g&1)) and (i, 2(g&1)+1) in their register half g>>1;
//  * one K-step = 128 bytes per row = the SAME LDS image as a 64-deep bf16 step ([rows][128 B], 16-B chunk ^= row & 7,
//    filled by the same buffer_load ... lds stream); a fragment is chunks g and 4+g of its row: the two conflict-free
//    ds_read_b128 of the bf16 kernel's two half-steps;
//  * scales travel as [K/32][ld] bytes (k-block major): the four k-blocks of a step x 256 rows are four 256-byte
//    segments per operand, one buffer_load_dword ... lds each, and a lane fetches its byte with ds_read_u8;
//  * the matrix pipe needs the same operand BYTES per cycle as the bf16 kernel at twice the FLOPs, so tiles are as
//    large as there: 256x256 (128x64 per wave, one fragment set, A reloaded in halves) and 256x128 / 128x128
//    (64x64 per wave, two fragment sets).
#include "gemm_common.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4v;

// 4-byte-per-lane LDS-DMA (scale segments): 64 lanes x 4 B -> LDS at lds_addr + 4*lane
__device__ __forceinline__ void lds_dma4(unsigned lds_addr, __attribute__((ext_vector_type(4))) int rsrc, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", "m0");
}

// 32 operand bytes of lane (i, g) for the row `row` of a [rows][128 B] swizzled image: k = 16g .. 16g+15 (registers 0-3) and
// k = 64+16g .. 64+16g+15 (registers 4-7), i.e. 16-byte chunks g and 4+g -- the two chunk reads of the bf16 kernel's ks = 0, 1
__device__ __forceinline__ i32x8 lds_mx_frag(const char* tile, int row, int lane) {
  const int g = lane >> 4, sw = row & 7;
  const i32x4v lo = *(LDS_PTR(const i32x4v))(tile + row * 128 + ((g ^ sw) << 4));
  const i32x4v hi = *(LDS_PTR(const i32x4v))(tile + row * 128 + (((4 + g) ^ sw) << 4));
  return (i32x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int OUTF32, int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 ? 2 : 1)) void gemm_mx8_kernel(const GemmArgs p) {
  constexpr int NWAVE = WM * WN;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, SC_BYTES = 8 * 256, STAGE = A_BYTES + B_BYTES + SC_BYTES;
  constexpr int A_PW = A_BYTES / 1024 / NWAVE, B_PW = B_BYTES / 1024 / NWAVE, S_PW = 8 / NWAVE, LPT = A_PW + B_PW + S_PW;
  constexpr int MT = BM / WM / 16, NTL = BN / WN / 16;
  static_assert(A_PW * NWAVE * 1024 == A_BYTES && B_PW * NWAVE * 1024 == B_BYTES && S_PW * NWAVE == 8, "tile must split evenly over waves");
  static_assert(NS == 2, "two ring slots");
  static_assert(BM <= 256 && BN <= 256, "a scale segment holds 256 rows");
  static_assert(MT * NTL <= 16 && MT % 2 == 0, "64x64 wave tiles: one activation fragment set in halves, two weight sets");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  char* scratch = smem + NS * STAGE + wave * 2048;

  const int G = gridDim.x, nxcd = min(8, G), q = G / nxcd, r = G - q * nxcd;
  const int xcd = blockIdx.x % nxcd, lb = blockIdx.x / nxcd, nx = q + (xcd < r ? 1 : 0);
  const int before = xcd * q + min(xcd, r);
  const int tiles = p.tiles_m * p.tiles_n;
  const int end = (int)((long)tiles * (before + nx) / G);
  const int beg = (int)((long)tiles * before / G);
  const bool dyn = p.sched != nullptr;
  unsigned* const qctr = p.sched + xcd;
  int* const qslot = (int*)(smem + NS * STAGE);
  auto leave = [&]() {
    if (dyn && tid == 0) {
      const unsigned done = atomicAdd(p.sched + 8, 1u);
      if (done == gridDim.x - 1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) __hip_atomic_store(p.sched + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  int wg, nwg;
  if (dyn) {
    if (tid == 0) {
      const int first = (int)atomicAdd(qctr, 2u);
      qslot[0] = beg + first;
      qslot[1] = beg + first + 1;
    }
    __syncthreads();
    wg = __builtin_amdgcn_readfirstlane(qslot[0]);
    nwg = __builtin_amdgcn_readfirstlane(qslot[1]);
    __syncthreads();
  } else {
    wg = beg + lb;
    nwg = wg + nx;
  }
  if (wg >= end) { leave(); return; }

  // operands as "bf16 pairs": a 128-byte fp8 row step is byte-for-byte the 64-element bf16 row step of gemm.hip
  const long lda2 = p.lda / 2, ldb2 = p.ldb / 2;
  const LaneOff aoff = operand_lane<0, BM, 64>(lda2, lane), boff = operand_lane<0, BN, 64>(ldb2, lane);
  const int nk = p.K / 128;

  auto decode = [&](int w) {
    TileInfo t;
    t.slice = 0;
    const int tm = w / p.tiles_n, tn = w - tm * p.tiles_n;
    t.m0 = tm * BM; t.n0 = tn * BN;
    t.kext = p.K; t.nk = nk;
    t.abase = (const bf16_t*)((const uint8_t*)p.A + (long)t.m0 * p.lda);
    t.bbase = (const bf16_t*)((const uint8_t*)p.B + (long)t.n0 * p.ldb);
    t.a_valid = p.M - t.m0; t.b_valid = p.N - t.n0;
    return t;
  };
  constexpr bool REBUILD = false;
  auto stage = [&](int s, const TileInfo& t, int kstep) {
    char* sa = smem + s * STAGE;
    if (REBUILD) {
      int l2 = lane;
      asm volatile("" : "+v"(l2));
      const LaneOff ao = operand_lane<0, BM, 64>(lda2, l2), bo = operand_lane<0, BN, 64>(ldb2, l2);
      stage_operand<0, BM, A_PW, 64>(sa, t.abase + kstep * 64, ao, lda2, t.a_valid, 64, wave, lane);
      stage_operand<0, BN, B_PW, 64>(sa + A_BYTES, t.bbase + kstep * 64, bo, ldb2, t.b_valid, 64, wave, lane);
    } else {
      stage_operand<0, BM, A_PW, 64>(sa, t.abase + kstep * 64, aoff, lda2, t.a_valid, 64, wave, lane);
      stage_operand<0, BN, B_PW, 64>(sa + A_BYTES, t.bbase + kstep * 64, boff, ldb2, t.b_valid, 64, wave, lane);
    }
    // scale segments: 0..3 = A k-blocks 4*kstep .. +3 (rows m0 .. m0+BM-1), 4..7 = B.  Rows beyond the matrix edge read
    // the padding of the scale row (ld is a multiple of 256 by contract): they only reach accumulators that are never stored.
    typedef __attribute__((ext_vector_type(4))) int i32x4;
#pragma unroll
    for (int t2 = 0; t2 < S_PW; ++t2) {
      const int seg = wave * S_PW + t2;                     // wave-uniform
      const bool isb = seg >= 4;
      const uint8_t* src = isb ? p.SB + (long)(kstep * 4 + seg - 4) * p.ldsb + t.n0 : p.SA + (long)(kstep * 4 + seg) * p.ldsa + t.m0;
      const unsigned long sb_ = (unsigned long)src;
      const i32x4 rs = {(int)(unsigned)sb_, (int)((sb_ >> 32) & 0xffffu), (int)OOB_OFF, 0x00020000};
      const int rows = isb ? BN : BM;
      const unsigned vo = (lane * 4 < rows) ? (unsigned)(lane * 4) : OOB_OFF;
      lds_dma4((unsigned)(unsigned long)(LDS_PTR(char))(sa + A_BYTES + B_BYTES + seg * 256), rs, vo, 0);
    }
  };

  f32x4 acc[MT][NTL];
  // Fragment registers (round 4: no scratch in the K loop).  Round 2 kept TWO full fragment sets (2 x 72 registers beside 64
  // accumulators): at the 256-register limit of an 8-wave block hipcc spilled 8-57 VGPRs and dozens of SGPRs, reloaded inside
  // the K loop (`make check-isa`).  Now ONE set of activation fragments (a[MT]) whose halves are refilled as soon as their
  // MFMA cluster has consumed them, and two sets of weight fragments (b / nb) that swap roles every step: 32 + 2 x 32 + scales
  // = ~110 registers beside the accumulators.
  struct AFr { i32x8 a[MT]; int sa[MT]; };
  struct BFr { i32x8 b[NTL]; int sb[NTL]; };
  auto load_a = [&](const char* st, AFr& f, int lo, int hi) {
    const int g = lane >> 4, i = lane & 15;
    const char* sc = st + A_BYTES + B_BYTES;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
      if (mi >= lo && mi < hi) {
        const int row = wm * MT * 16 + mi * 16 + i;
        f.a[mi] = lds_mx_frag(st, row, lane);
        f.sa[mi] = (int)*(LDS_PTR(const uint8_t))(sc + g * 256 + row);
      }
  };
  auto load_b = [&](const char* st, BFr& f) {
    const int g = lane >> 4, i = lane & 15;
    const char* sc = st + A_BYTES + B_BYTES;
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni) {
      const int row = wn * NTL * 16 + ni * 16 + i;
      f.b[ni] = lds_mx_frag(st + A_BYTES, row, lane);
      f.sb[ni] = (int)*(LDS_PTR(const uint8_t))(sc + (4 + g) * 256 + row);
    }
  };
  auto mma_rows = [&](const AFr& fa, const BFr& fb, int lo, int hi) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
      if (mi >= lo && mi < hi) {
#pragma unroll
        for (int ni = 0; ni < NTL; ++ni)     // transposed product: the weight fragment is the A operand, so a lane owns 4 columns of a row
          acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb.b[ni], fa.a[mi], acc[mi][ni], 0, 0, 0, fb.sb[ni], 0, fa.sa[mi]);
      }
  };

  // The operand stream is one sequence of 128-deep steps over (tile, K-step), every tile padded to an EVEN number of
  // steps (a padding step is staged with every lane out of range: zero operands, zero scale bytes, adds nothing), so the
  // two weight-fragment sets keep fixed roles: even steps multiply set B and fill set NB, odd steps the other way round, and a
  // tile always ends with the NEXT tile's first fragments in place.  Every step is the same unconditional code:
  //   DMA(position s+1) | read a[MT/2..MT) of s | MFMA(rows 0..MT/2 of s) | retire s+1, barrier |
  //   read a[0..MT/2) and the weight fragments of s+1 | MFMA(rows MT/2..MT of s)
  // The upper activation half of position s is read at the top of step s (its registers were freed by the last cluster of
  // step s-1) and lands under the first cluster; the lower half and the weights of s+1 land under the second.  All reads of a
  // stage happen between the barrier that publishes it and the barrier of the step after, i.e. before its slot is restaged:
  // position s sits in slot s & 1, is restaged (position s+2) at the top of step s+1 -- after the barrier in the middle of step
  // s, which every wave passes only with its upper-half reads of position s retired (lgkmcnt(0) in front of it).
  const int nkp = (nk + 1) & ~1;
  TileInfo T = decode(wg);
  auto stage_pos = [&](int slot, const TileInfo& t, int ks) {
    if (ks < nk) { stage(slot, t, ks); return; }
    // padding step: same number of DMA instructions (vmcnt bookkeeping), every lane out of range -> zeros
    char* sa = smem + slot * STAGE;
    const LaneOff oob = {OOB_OFF, 0};
    stage_operand<0, BM, A_PW, 64>(sa, t.abase, oob, lda2, 0, 0, wave, lane);
    stage_operand<0, BN, B_PW, 64>(sa + A_BYTES, t.bbase, oob, ldb2, 0, 0, wave, lane);
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    const unsigned long sb_ = (unsigned long)p.SA;
    const i32x4 rs = {(int)(unsigned)sb_, (int)((sb_ >> 32) & 0xffffu), (int)OOB_OFF, 0x00020000};
#pragma unroll
    for (int t2 = 0; t2 < S_PW; ++t2)
      lds_dma4((unsigned)(unsigned long)(LDS_PTR(char))(sa + A_BYTES + B_BYTES + (wave * S_PW + t2) * 256), rs, OOB_OFF, 0);
  };
  stage_pos(0, T, 0);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  int cur = 0, nxt = 1;
  bool parked = false;
  AFr fa;
  BFr fbA, fbB;
  load_a(smem, fa, 0, MT / 2);
  load_b(smem, fbA);
  for (;;) {
    if (parked) {
      __builtin_amdgcn_s_barrier();     // the id parked by wave 0 after its last epilogue is visible
      nwg = __builtin_amdgcn_readfirstlane(qslot[0]);
    }
    const bool has_next = nwg < end;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NTL; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto kstep = [&](int s, BFr& fc, BFr& fn) {
      if (s + 1 < nkp) stage_pos(nxt, T, s + 1);
      else if (has_next) { const TileInfo NX = decode(nwg); stage_pos(nxt, NX, 0); }
      __builtin_amdgcn_sched_barrier(0);
      load_a(smem + cur * STAGE, fa, MT / 2, MT);   // upper half of THIS position: lands under the first cluster
      __builtin_amdgcn_s_setprio(1);
      mma_rows(fa, fc, 0, MT / 2);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      cur ^= 1; nxt ^= 1;
      wait_vmcnt<0>();                              // this wave's share of position s+1 has landed ...
      __builtin_amdgcn_s_waitcnt(0xC07F);           // ... its reads of position s are complete (slot may be restaged) ...
      __builtin_amdgcn_s_barrier();                 // ... and everybody's
      __builtin_amdgcn_sched_barrier(0);
      load_a(smem + cur * STAGE, fa, 0, MT / 2);    // position s+1: lower activation half (its registers are free) + weights;
      load_b(smem + cur * STAGE, fn);               // land under the second cluster (garbage, never used, after the last position)
      __builtin_amdgcn_s_setprio(1);
      mma_rows(fa, fc, MT / 2, MT);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < nkp; s += 2) { kstep(s, fbA, fbB); kstep(s + 1, fbB, fbA); }
    int drawn = end;
    if (dyn && has_next && tid == 0) drawn = beg + (int)atomicAdd(qctr, 1u);
    if (OUTF32) {
      store_tile<1, 0, MT, NTL>(acc, p, T, scratch, wm, wn, lane);
    } else {
      switch (p.epi) {
        case 1: store_tile<0, 1, MT, NTL>(acc, p, T, scratch, wm, wn, lane); break;
        case 2: store_tile<0, 2, MT, NTL>(acc, p, T, scratch, wm, wn, lane); break;
        case 3: store_tile<0, 3, MT, NTL>(acc, p, T, scratch, wm, wn, lane); break;
        default: store_tile<0, 0, MT, NTL>(acc, p, T, scratch, wm, wn, lane); break;
      }
    }
    if (!has_next) break;
    T = decode(nwg);
    wg = nwg;
    if (dyn) {
      if (tid == 0) qslot[0] = drawn;
      __builtin_amdgcn_s_waitcnt(0xC07F);
      parked = true;
    } else {
      nwg = wg + nx;
    }
  }
  leave();
}

// ------------------------------------------------------------------------------------------------ loader / consumer form
// gemm.hip's config 9 for MX-fp8 operands (round 4): 256x128 tile, 8 consumer waves + 4 loader waves, and -- because the loaders'
// stages no longer carry the scale segments -- a ring of THREE 48 KB slots where gemm_mx8_kernel has two (its 2 KB of scales per
// stage put a third slot 6 KB over the CU's LDS): the operand stream runs two 128-deep steps ahead instead of being drained
// (`vmcnt(0)`) at every step.  The consumers fetch their scale bytes straight from global memory (read-only inputs, no ring
// needed: 8 one-byte loads per lane and step, issued one step ahead, the compiler's own vmcnt bookkeeping since consumers issue
// no LDS-DMA).  Protocol and barrier count as gemm_ws_kernel.
template <int OUTF32>
__global__ __launch_bounds__(768) void gemm_mx8_ws_kernel(const GemmArgs p) {
  constexpr int BM = 256, BN = 128, NS = 3, WN = 2, NCW = 8, NLW = 4, MT = 4, NTL = 4;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PL = A_BYTES / 1024 / NLW, B_PL = B_BYTES / 1024 / NLW, LPL = A_PL + B_PL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x, nxcd = min(8, G), q = G / nxcd, r = G - q * nxcd;
  const int xcd = blockIdx.x % nxcd, lb = blockIdx.x / nxcd, nx = q + (xcd < r ? 1 : 0);
  const int before = xcd * q + min(xcd, r);
  const int tiles = p.tiles_m * p.tiles_n;
  const int end = (int)((long)tiles * (before + nx) / G), beg = (int)((long)tiles * before / G);
  // work queue as gemm_ws_kernel (gemm.hip): first unit static, the rest drawn by consumer wave 0 and handed to the other eleven
  // waves through the first word of its idle epilogue scratch behind the barrier of the tile's last-but-one position
  const bool dyn = p.sched != nullptr;
  unsigned* const qctr = p.sched + xcd;
  int* const qslot = (int*)(smem + NS * STAGE);
  const int wg0 = beg + lb, wg1 = wg0 + nx;
  auto leave = [&]() {
    if (dyn && tid == 0) {
      const unsigned done = atomicAdd(p.sched + 8, 1u);
      if (done == gridDim.x - 1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) __hip_atomic_store(p.sched + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  if (wg0 >= end) { leave(); return; }
  const int nk = p.K / 128;                           // >= 2 when the queue is used (host)
  const long lda2 = p.lda / 2, ldb2 = p.ldb / 2;      // operands as "bf16 pairs": a 128-byte fp8 row step = a 64-element bf16 one
  if (wave >= NCW) {
    // ------------------------------------------------------------------ loader wave
    __builtin_amdgcn_s_setprio(3);
    const int lw = wave - NCW;
    const LaneOff aoff = operand_lane<0, BM, 64>(lda2, lane), boff = operand_lane<0, BN, 64>(ldb2, lane);
    int iw = wg0, ik = 0, islot = 0;
    int known_next = dyn ? end : wg1;
    bool more = true, crossed = false;
    auto issue_next = [&]() -> bool {
      if (crossed) {
        crossed = false;
        iw = known_next;
        if (iw >= end) more = false;
      }
      if (!more) return false;
      const int tm = iw / p.tiles_n, tn = iw - tm * p.tiles_n;
      const bf16_t* abase = (const bf16_t*)((const uint8_t*)p.A + (long)tm * BM * p.lda);
      const bf16_t* bbase = (const bf16_t*)((const uint8_t*)p.B + (long)tn * BN * p.ldb);
      char* sa = smem + islot * STAGE;
      stage_operand<0, BM, A_PL, 64>(sa, abase + ik * 64, aoff, lda2, p.M - tm * BM, 64, lw, lane);
      stage_operand<0, BN, B_PL, 64>(sa + A_BYTES, bbase + ik * 64, boff, ldb2, p.N - tn * BN, 64, lw, lane);
      islot = (islot + 1 == NS) ? 0 : islot + 1;
      if (++ik == nk) { ik = 0; crossed = true; }
      return true;
    };
    int pending = 0;
    if (issue_next()) ++pending;
    if (issue_next()) ++pending;
    for (int w = wg0; w < end;) {
      for (int it = 0; it < nk; ++it) {
        if (pending >= 2) wait_vmcnt<LPL>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (dyn && it == nk - 2) {
          const int v = *(LDS_PTR(const int))qslot;
          __builtin_amdgcn_s_waitcnt(0xC07F);
          known_next = __builtin_amdgcn_readfirstlane(v);
        }
        --pending;
        if (issue_next()) ++pending;
      }
      w = known_next;
      if (!dyn) known_next = w + nx;
    }
    return;
  }
  // -------------------------------------------------------------------- consumer wave
  const int wm = wave / WN, wn = wave % WN, g = lane >> 4, i = lane & 15;
  char* scratch = smem + NS * STAGE + wave * 2048;
  f32x4 acc[MT][NTL];
  int cur = 0;
  const bool drawer = dyn && wave == 0;
  int drawn = end, nextv = dyn ? end : wg1;
  if (drawer && lane == 0) drawn = (int)atomicAdd(qctr, 1u);
  for (int w = wg0; w < end;) {
    TileInfo T;
    T.slice = 0;
    const int tm = w / p.tiles_n, tn = w - tm * p.tiles_n;
    T.m0 = tm * BM; T.n0 = tn * BN; T.kext = p.K; T.nk = nk;
    T.abase = nullptr; T.bbase = nullptr; T.a_valid = p.M - T.m0; T.b_valid = p.N - T.n0;
    // scale bytes of lane (i, g): row (wave tile row + 16 mi + i), k-block 4 ks + g of [K/32][ld] (rows past the matrix edge
    // read the padding of the scale row: ld is a multiple of 256 by contract)
    const uint8_t* sap = p.SA + (long)g * p.ldsa + T.m0 + wm * 64 + i;
    const uint8_t* sbp = p.SB + (long)g * p.ldsb + T.n0 + wn * 64 + i;
    int sa_c[MT], sb_c[NTL], sa_n[MT], sb_n[NTL];
    auto load_scales = [&](int ks, int (&sa_)[MT], int (&sb_)[NTL]) {
      const int kc = ks < nk ? ks : nk - 1;                 // (past the last step: a valid address, the values are not used)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) sa_[mi] = (int)sap[(long)kc * 4 * p.ldsa + mi * 16];
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni) sb_[ni] = (int)sbp[(long)kc * 4 * p.ldsb + ni * 16];
    };
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NTL; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    load_scales(0, sa_c, sb_c);
    auto kstep = [&](int it, int (&sa_)[MT], int (&sb_)[NTL], int (&san)[MT], int (&sbn)[NTL]) {
      if (drawer && it == nk - 2 && lane == 0) qslot[0] = beg + nx + drawn;
      __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's reads of the previous position are in registers
      __builtin_amdgcn_s_barrier();
      if (dyn && it == nk - 2) nextv = *(LDS_PTR(const int))qslot;
      load_scales(it + 1, san, sbn);           // next step's scales: in flight under this step's MFMAs
      const char* st = smem + cur * STAGE;
      i32x8 fa[MT], fb[NTL];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) fa[mi] = lds_mx_frag(st, wm * 64 + mi * 16 + i, lane);
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni) fb[ni] = lds_mx_frag(st + A_BYTES, wn * 64 + ni * 16 + i, lane);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTL; ++ni)         // transposed product: the weight fragment is the A operand, so a lane owns 4 columns of a row
          acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0, sb_[ni], 0, sa_[mi]);
      cur = (cur + 1 == NS) ? 0 : cur + 1;
    };
    for (int it = 0; it < nk; it += 2) {
      kstep(it, sa_c, sb_c, sa_n, sb_n);
      if (it + 1 < nk) kstep(it + 1, sa_n, sb_n, sa_c, sb_c);
      else {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) sa_c[mi] = sa_n[mi];
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);        // (the read of the next id included)
    if (drawer && lane == 0) drawn = (int)atomicAdd(qctr, 1u);     // the id of the tile after next, under the epilogue
    if (OUTF32) {
      store_tile<1, 0, MT, NTL>(acc, p, T, scratch, wm, wn, lane);
    } else {
      switch (p.epi) {
        case 1: store_tile<0, 1, MT, NTL>(acc, p, T, scratch, wm, wn, lane); break;
        case 2: store_tile<0, 2, MT, NTL>(acc, p, T, scratch, wm, wn, lane); break;
        case 3: store_tile<0, 3, MT, NTL>(acc, p, T, scratch, wm, wn, lane); break;
        default: store_tile<0, 0, MT, NTL>(acc, p, T, scratch, wm, wn, lane); break;
      }
    }
    w = __builtin_amdgcn_readfirstlane(nextv);
    nextv = dyn ? end : w + nx;
  }
  leave();
}

// ------------------------------------------------------------------------------------------------ quantiser
// X[M][K] (bf16 or fp32) -> Xq[M][K] e4m3 bytes + S[K/32][lds] E8M0 bytes.  Per 32-element block: scale 2^e with
// e = ceil(log2(amax / 448)) (the smallest power of two that brings the block inside e4m3's finite range, so nothing
// saturates), elements = round-to-nearest-even(x * 2^-e).  amax == 0 -> e = -127 (byte 0), elements 0.
// One wave per row, every 16-byte chunk of the row in flight at once (NCH chunks per lane); a 32-block is 4 neighbouring
// lanes (two xor-shuffles).  The same block code is used by the fused LayerNorm output (rowops.hip).
template <typename TIN, int NCH>
__global__ __launch_bounds__(256) void quant_mx8_kernel(const TIN* __restrict__ X, long ldx, int M, int K, uint8_t* __restrict__ Q, long ldq,
                                                        uint8_t* __restrict__ S, long lds_) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = K >> 3;
  for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
    float f[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        if constexpr (sizeof(TIN) == 2) {
          bf8_to_f32(*(const bf16x8*)((const bf16_t*)X + row * ldx + ch * 8), f[c]);
        } else {
          const f32x4 a = *(const f32x4*)((const float*)X + row * ldx + ch * 8), b = *(const f32x4*)((const float*)X + row * ldx + ch * 8 + 4);
          f[c][0] = a[0]; f[c][1] = a[1]; f[c][2] = a[2]; f[c][3] = a[3]; f[c][4] = b[0]; f[c][5] = b[1]; f[c][6] = b[2]; f[c][7] = b[3];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[c][j] = 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      typedef __attribute__((ext_vector_type(2))) int i32x2;
      int e;
      const i32x2 w = mx8_block(f[c], e);
      if (ch < nchunk) {
        *(i32x2*)(Q + row * ldq + ch * 8) = w;
        if ((lane & 3) == 0) S[(long)(ch >> 2) * lds_ + row] = (uint8_t)e;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ host
struct MxCfg { int bm, bn, wm, wn; };
static const MxCfg kMx[] = {
    {256, 128, 4, 2},     // 0: 64x64 per wave, 8 waves, one block per CU
    {128, 128, 2, 2},     // 1: 64x64 per wave, 4 waves, two blocks per CU
    {256, 128, 4, 2},     // 2: config 0's tile on 8 consumer + 4 loader waves, three-slot ring (gemm_mx8_ws_kernel)
};
static int g_force_mx = -1;
extern "C" int hriemo_gemm_mx8_force_config(int cfg) {
  g_force_mx = (cfg >= 0 && cfg < 3) ? cfg : -1;
  return 0;
}

template <int OUTF32, int BM, int BN, int WM, int WN>
static void launch_mx(const GemmArgs& a, hipStream_t st) {
  constexpr int NS = 2;
  const int lds = NS * (BM * 128 + BN * 128 + 2048) + WM * WN * 2048;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_mx8_kernel<OUTF32, BM, BN, WM, WN, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const long units = (long)a.tiles_m * a.tiles_n;
  const long slots = (long)hriemo_num_cus() * (lds <= 80 * 1024 ? 2 : 1);
  const int grid = (int)(units < slots ? units : slots);
  GemmArgs q = a;
  q.sched = units > slots ? hriemo_gemm_sched_slot(st) : nullptr;
  hipLaunchKernelGGL((gemm_mx8_kernel<OUTF32, BM, BN, WM, WN, NS>), dim3(grid), dim3(WM * WN * 64), lds, st, q);
}

template <int OUTF32>
static void launch_mx_ws(const GemmArgs& a, hipStream_t st) {
  const int lds = 3 * (256 * 128 + 128 * 128) + 8 * 2048;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_mx8_ws_kernel<OUTF32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const long units = (long)a.tiles_m * a.tiles_n, slots = hriemo_num_cus();
  GemmArgs q = a;
  q.sched = (units > slots && a.K >= 256 && !(hriemo_gemm_debug_flags_get() & 8)) ? hriemo_gemm_sched_slot(st) : nullptr;   // (>= 2 steps per unit)
  hipLaunchKernelGGL((gemm_mx8_ws_kernel<OUTF32>), dim3((int)(units < slots ? units : slots)), dim3(768), lds, st, q);
}

static int gemm_mx8_impl(int M, int N, int K, const void* Aq, long lda, const void* SA, long ldsa, const void* Bq, long ldb,
                         const void* SB, long ldsb, void* C, long ldc, int c_is_f32, const float* bias, int epilogue,
                         const void* aux, long ldaux, void* CQ, long ldcq, void* SC, long ldsc, hipStream_t st) {
  HRIEMO_CHECK(M > 0 && N > 0 && K > 0, "gemm_mx8: empty problem M=%d N=%d K=%d", M, N, K);
  HRIEMO_CHECK(K % 128 == 0, "gemm_mx8: K=%d must be a multiple of 128 (one scaled MFMA step)", K);
  HRIEMO_CHECK(N % 8 == 0, "gemm_mx8: N=%d must be a multiple of 8", N);
  HRIEMO_CHECK(lda % 16 == 0 && ldb % 16 == 0 && ldc % (c_is_f32 ? 4 : 8) == 0, "gemm_mx8: leading dims must keep 16-byte alignment");
  HRIEMO_CHECK(ldsa % 256 == 0 && ldsb % 256 == 0 && ldsa >= M && ldsb >= N, "gemm_mx8: scale rows must be padded to a multiple of 256 (ldsa=%ld ldsb=%ld)", ldsa, ldsb);
  HRIEMO_CHECK(((uintptr_t)Aq % 16) == 0 && ((uintptr_t)Bq % 16) == 0 && ((uintptr_t)C % 16) == 0 && ((uintptr_t)SA % 4) == 0 && ((uintptr_t)SB % 4) == 0, "gemm_mx8: unaligned operand");
  HRIEMO_CHECK(lda < (1L << 22) && ldb < (1L << 22), "gemm_mx8: leading dimension too large for 32-bit tile offsets");
  HRIEMO_CHECK(c_is_f32 || (epilogue >= 0 && epilogue <= 3), "gemm_mx8: bad epilogue");
  HRIEMO_CHECK(epilogue < 2 || (aux != nullptr && ldaux % 8 == 0 && ((uintptr_t)aux % 16) == 0), "gemm_mx8: epilogue 2/3 needs a 16-byte aligned aux");
  HRIEMO_CHECK(!(c_is_f32 && epilogue != 0), "gemm_mx8: fp32 output has no activation epilogue");
  int cfg = g_force_mx;
  if (cfg < 0) cfg = (M < 1024 || N < 256) ? 1 : 2;
  GemmArgs a = {};
  a.M = M; a.N = N; a.K = K;
  a.A = (const bf16_t*)Aq; a.lda = lda; a.B = (const bf16_t*)Bq; a.ldb = ldb;
  a.SA = (const uint8_t*)SA; a.ldsa = ldsa; a.SB = (const uint8_t*)SB; a.ldsb = ldsb;
  a.C = C; a.ldc = ldc; a.bias = bias; a.aux = (const bf16_t*)aux; a.ldaux = ldaux; a.epi = epilogue;
  a.tiles_m = (M + kMx[cfg].bm - 1) / kMx[cfg].bm; a.tiles_n = (N + kMx[cfg].bn - 1) / kMx[cfg].bn;
  a.splitk = 1; a.k_per_split = K; a.ws = nullptr; a.accumulate = 0; a.sched = nullptr;
  if (CQ != nullptr) {
    HRIEMO_CHECK(!c_is_f32 && epilogue <= 1 && N % 32 == 0 && SC != nullptr && ldcq % 8 == 0 && ((uintptr_t)CQ % 8) == 0 && ldsc >= M,
                 "gemm_mx8: the fused MX copy of the output needs a bf16 output, epilogue 0 / 1, N %% 32 == 0 and a scale buffer");
    a.CQ = (uint8_t*)CQ; a.ldcq = ldcq; a.SC = (uint8_t*)SC; a.ldsc = ldsc;
  }
  hriemo_prof_begin(HP_GEMM_MX8, st);
  if (cfg == 2) {
    if (c_is_f32) launch_mx_ws<1>(a, st); else launch_mx_ws<0>(a, st);
  } else if (c_is_f32) {
    if (cfg == 0) launch_mx<1, 256, 128, 4, 2>(a, st); else launch_mx<1, 128, 128, 2, 2>(a, st);
  } else {
    if (cfg == 0) launch_mx<0, 256, 128, 4, 2>(a, st); else launch_mx<0, 128, 128, 2, 2>(a, st);
  }
  HRIEMO_LAUNCH_CHECK("gemm_mx8_kernel");
  hriemo_prof_end(HP_GEMM_MX8, st, 2.0 * M * N * K);
  return 0;
}

extern "C" int hriemo_gemm_mx8(int M, int N, int K, const void* Aq, long lda, const void* SA, long ldsa, const void* Bq, long ldb,
                               const void* SB, long ldsb, void* C, long ldc, int c_is_f32, const float* bias, int epilogue,
                               const void* aux, long ldaux, hipStream_t st) {
  return gemm_mx8_impl(M, N, K, Aq, lda, SA, ldsa, Bq, ldb, SB, ldsb, C, ldc, c_is_f32, bias, epilogue, aux, ldaux, nullptr, 0, nullptr, 0, st);
}
// same, and the epilogue also leaves the MX-fp8 form of the bf16 output it stores -- bytes CQ[M][ldcq], scales SC[N/32][ldsc] -- for
// the GEMM that reads this output next (FFN1's ReLU output as FFN2's operand): no separate quantisation pass over [M, N]
extern "C" int hriemo_gemm_mx8_q(int M, int N, int K, const void* Aq, long lda, const void* SA, long ldsa, const void* Bq, long ldb,
                                 const void* SB, long ldsb, void* C, long ldc, const float* bias, int epilogue, void* CQ, long ldcq,
                                 void* SC, long ldsc, hipStream_t st) {
  HRIEMO_CHECK(CQ != nullptr, "gemm_mx8_q: CQ required");
  return gemm_mx8_impl(M, N, K, Aq, lda, SA, ldsa, Bq, ldb, SB, ldsb, C, ldc, 0, bias, epilogue, nullptr, 0, CQ, ldcq, SC, ldsc, st);
}

extern "C" long hriemo_mx8_scale_ld(int rows) { return ((long)rows + 255) / 256 * 256; }

extern "C" int hriemo_quant_mx8(const void* X, long ldx, int src_is_f32, int M, int K, void* Xq, long ldq, void* S, long lds_,
                                hipStream_t st) {
  HRIEMO_CHECK(M > 0 && K > 0 && K % 32 == 0 && K <= 4096, "quant_mx8: K=%d must be a positive multiple of 32 (<= 4096)", K);
  HRIEMO_CHECK(ldx % 8 == 0 && ldq % 8 == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Xq % 8) == 0, "quant_mx8: unaligned operand");
  HRIEMO_CHECK(lds_ % 256 == 0 && lds_ >= M && ((uintptr_t)S % 4) == 0, "quant_mx8: scale rows must be padded to a multiple of 256");
  const int nch = (K / 8 + 63) / 64;
  int grid = (M + 3) / 4;
  if (grid > 4096) grid = 4096;
  hriemo_prof_begin(HP_ROWOPS, st);
#define QCALL(T, N) hipLaunchKernelGGL((quant_mx8_kernel<T, N>), dim3(grid), dim3(256), 0, st, (const T*)X, ldx, M, K, (uint8_t*)Xq, ldq, (uint8_t*)S, lds_)
#define QDISP(T)                               \
  if (nch <= 1) { QCALL(T, 1); }               \
  else if (nch <= 2) { QCALL(T, 2); }          \
  else if (nch <= 4) { QCALL(T, 4); }          \
  else { QCALL(T, 8); }
  if (src_is_f32) { QDISP(float) } else { QDISP(bf16_t) }
#undef QDISP
#undef QCALL
  HRIEMO_LAUNCH_CHECK("quant_mx8_kernel");
  hriemo_prof_end(HP_ROWOPS, st, (double)M * K * (src_is_f32 ? 5 : 3) + (double)M * K / 32);
  return 0;
}
