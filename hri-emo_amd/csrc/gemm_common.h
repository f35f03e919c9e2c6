// Shared device pieces of the GEMM kernels (gemm.hip: bf16 operands; gemm_mx8.hip: MX-fp8 operands):
// launch arguments, LDS fragment reads, LDS-DMA operand staging, work-unit decoding, the fused epilogue.
#pragma once
#include "common.h"
#include <mutex>
#include <type_traits>

struct GemmArgs {
  int M, N, K;
  const bf16_t* A; long lda;
  const bf16_t* B; long ldb;
  void* C; long ldc;
  const float* bias;
  const bf16_t* aux; long ldaux;
  int epi;  // 0 none, 1 relu, 2 multiply by (aux > 0), 3 add aux
  int tiles_m, tiles_n, splitk, k_per_split;
  float* ws;
  int accumulate;
  unsigned* sched;   // per-launch work queue: [0..7] per-XCD next-unit counters, [8] blocks finished; nullptr = static walk
  int flags;         // bit 0: the first K-step after an epilogue counts that epilogue's stores in its retire wait (gemm.hip)
  // optional column sums of the stored tile (EPI 2 only: the ReLU-masked dX of an FFN, whose column sums are the first Linear's
  // bias gradient): one fp32 partial row per wave row-block, [ceil(M / (wave tile rows))][N]; nullptr = none
  float* cs;
  // MX-fp8 operands only (gemm_mx8.hip): E8M0 block scales, one byte per 32 k-elements, laid out [K/32][ld] (k-block major)
  const uint8_t* SA; long ldsa;
  const uint8_t* SB; long ldsb;
  // optional fused re-quantisation of the bf16 output tile for the next GEMM: fp8 bytes [M][ldcq] + scales [N/32][ldsc]
  uint8_t* CQ; long ldcq;
  uint8_t* SC; long ldsc;
};

// host helpers defined in gemm.hip
unsigned* hriemo_gemm_sched_slot(hipStream_t st);   // work-queue words of the persistent kernels, one slot per (device, stream)
int hriemo_num_cus();
int hriemo_gemm_debug_flags_get();                 // current value of the tuning word set by hriemo_gemm_debug_flags

// chunk swizzle of the 64-byte-row image (BK = 32): rows r..r+3 share one 256-B bank row, so the 16-B chunk c of
// row r is stored at c ^ h((r>>2)&3), h = {0,2,3,1}: conflict-free for the ds_read_b128 lane groups
__device__ __forceinline__ int swz4(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

template <int BK>
__device__ __forceinline__ bf16x8 lds_row_frag(const char* tile, int sub0, int ks, int lane) {
  const int row = sub0 + (lane & 15);
  if (BK == 64) {
    const int c = ks * 4 + (lane >> 4);
    return *(LDS_PTR(const bf16x8))(tile + row * 128 + ((c ^ (row & 7)) << 4));
  } else {
    const int c = lane >> 4;
    return *(LDS_PTR(const bf16x8))(tile + row * 64 + ((c ^ swz4(row)) << 4));
  }
}

template <int ROWB>   // bytes per k-row of the strided image (256 or 512)
__device__ __forceinline__ bf16x8 lds_tr_frag(const char* tile, int sub0, int ks, int lane) {
  const int g = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3;
  const int kr = ks * 32 + 8 * g + qq;
  const int key = qq | ((g & 1) << 2);
  const int c = ((sub0 >> 3) ^ (key << 1)) | (pp >> 1);
  const char* a0 = tile + kr * ROWB + c * 16 + (pp & 1) * 8;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(a0 + 4 * ROWB));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}

// ---- operand staging through buffer_load ... lds -------------------------------------------------
// Per lane and wave-instruction the byte offset inside a tile's operand window is the same for every tile
// and K-step; only the (wave-uniform) descriptor base moves.  So the steady-state loop spends no VALU on
// addresses.  Edge tiles and the ragged last K-step take a slower path that rebuilds the lane's row / column
// and replaces out-of-matrix lanes by offset 0x80000000 >= num_records, which the hardware range check turns
// into zeros (no branches around the loads, no clamping).
#define OOB_OFF 0x80000000u

// Lane-invariant part of the operand addressing.  Wave-instruction j of an operand covers
//   K-contiguous (T == 0): rows 8j..8j+7, lane -> row 8j + lane/8, 16-B chunk (lane&7) ^ (row&7); row&7 does not
//     depend on j, so ONE per-lane byte offset serves every j and the j-dependent part is a scalar (soffset);
//   K-strided (T == 1): k-rows RPI*j .. +RPI-1, lane -> k-row RPI*j + lane/CPR, chunk (lane%CPR) ^ (key(kr)<<1);
//     key splits into a lane part (lane/CPR) and a wave-uniform part (from RPI*j), so two per-lane values plus
//     one v_xor/v_add per instruction rebuild the offset.
struct LaneOff { unsigned a, b; };

// One LDS-DMA wave-instruction: 64 lanes x 16 bytes, global (descriptor + per-lane offset + scalar offset) ->
// LDS at lds_addr + 16*lane.  Issued through inline asm on purpose: hipcc tracks the builtin form as an LDS store
// that may alias every later ds_read and, once the K loop sits inside the persistent tile loop, puts
// s_waitcnt vmcnt(0) between the DMA issue and the fragment reads of the same iteration, which serialises the
// whole prefetch ring.  Hidden from the compiler, the ring is ordered by this file's own counted vmcnt waits and
// barriers alone; waits the compiler computes for its own loads only become more conservative (the hidden
// operations are younger than anything it waits for, and vmcnt retires in order).
__device__ __forceinline__ void lds_dma16(unsigned lds_addr, __attribute__((ext_vector_type(4))) int rsrc, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", "m0");
}

template <int T, int ROWS, int BK>
__device__ __forceinline__ LaneOff operand_lane(long ld, int lane) {
  LaneOff lo;
  if (T == 0) {
    if (BK == 64) {
      const int row = lane >> 3;
      lo.a = (unsigned)((row * ld + ((lane & 7) ^ row) * 8) * 2);
    } else {                             // 64-byte rows: 16 rows per instruction, chunk (lane&3) ^ swz4(row)
      const int row = lane >> 2;
      lo.a = (unsigned)((row * ld + ((lane & 3) ^ swz4(row)) * 8) * 2);
    }
    lo.b = 0;
  } else {
    constexpr int CPR = ROWS / 8;        // 16-B chunks per k-row
    const int lr = lane / CPR, lc = lane % CPR;
    lo.a = (unsigned)(lr * ld * 2);
    lo.b = (unsigned)((lc ^ (lr << 1)) << 4);
  }
  return lo;
}

// valid = rows (T == 0) or columns (T == 1) of this tile that lie inside the matrix; krem = valid k extent of
// this K-step (>= BK in the steady state)
template <int T, int ROWS, int PW, int BK>
__device__ __forceinline__ void stage_operand(char* tile, const bf16_t* kbase, const LaneOff lo, long ld, int valid, int krem, int wave, int lane) {
  // raw buffer descriptor {base[31:0], base[47:32] (stride 0), num_records, flags}, in SGPRs
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  const unsigned long kb = (unsigned long)kbase;
  const i32x4 rsrc = {(int)(unsigned)kb, (int)((kb >> 32) & 0xffffu), (int)OOB_OFF, 0x00020000};
  const bool full = valid >= ROWS && krem >= BK;     // wave-uniform
  constexpr int CPR = ROWS / 8;
  constexpr int RPI = 64 / CPR;          // k-rows per wave-instruction (T == 1)
  constexpr int RPJ = BK == 64 ? 8 : 16; // rows per wave-instruction (T == 0)
  if (full) {                            // interior tile, full K-step: no per-lane work at all
#pragma unroll
    for (int t = 0; t < PW; ++t) {
      const int j = wave * PW + t;
      if (T == 0) {
        lds_dma16((unsigned)(unsigned long)(LDS_PTR(char))(tile + j * 1024), rsrc, lo.a, (int)(j * RPJ * 2 * ld));
      } else {
        const int kr0 = j * RPI;
        const int ukey = (kr0 & 3) | (((kr0 >> 3) & 1) << 2);
        lds_dma16((unsigned)(unsigned long)(LDS_PTR(char))(tile + j * 1024), rsrc, lo.a + (lo.b ^ (unsigned)(ukey << 5)), (int)(kr0 * 2 * ld));
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < PW; ++t) {
    const int j = wave * PW + t;
    unsigned o;
    int so;
    if (T == 0) {
      o = lo.a;
      so = (int)(j * RPJ * 2 * ld);
      const int row = j * RPJ + (BK == 64 ? (lane >> 3) : (lane >> 2));
      const int c = BK == 64 ? ((lane & 7) ^ (row & 7)) : ((lane & 3) ^ swz4(row));
      if (row >= valid || c * 8 >= krem) o = OOB_OFF;
    } else {
      const int kr0 = j * RPI;
      const int ukey = (kr0 & 3) | (((kr0 >> 3) & 1) << 2);
      const unsigned cb = lo.b ^ (unsigned)(ukey << 5);      // 16 * chunk
      o = lo.a + cb;
      so = (int)(kr0 * 2 * ld);
      if ((int)(cb >> 1) >= valid || kr0 + lane / CPR >= krem) o = OOB_OFF;
    }
    lds_dma16((unsigned)(unsigned long)(LDS_PTR(char))(tile + j * 1024), rsrc, o, so);
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One unit of work of a block: an output tile (and, for split-K, one K slice of it).
struct TileInfo {
  const bf16_t* abase;
  const bf16_t* bbase;
  int m0, n0, slice, a_valid, b_valid, kext, nk;
};

// ---- epilogue.  A lane owns C[m = .. + (lane&15)][n = .. + 4*(lane>>4) .. +3] of each 16x16 accumulator tile.
// fp32 (weight gradients / split-K slabs): 16-byte stores straight from the accumulators.
// bf16: bias / residual in fp32 registers, ReLU on the packed result, then each wave transposes 16 rows at a
// time through its PRIVATE 2 KB of LDS (wave tiles are 64 columns = one 128-B line wide) so every global store
// is 16 bytes of a full line.  No block barrier and no use of the operand ring: the next tile's operands are
// already streaming into it.  The epilogue is instruction-bound (two waves per SIMD, ~MT*40 VALU/LDS ops each),
// so the variant is a compile-time parameter, addresses are one per-lane offset + scalar offsets of a buffer
// descriptor (edge lanes get an out-of-range offset: dropped stores / zero loads instead of branches), and all
// aux loads of a tile are issued before its first store (vmcnt retires in order: a load behind a store would
// wait for the store's acknowledgement).
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4v;
typedef __attribute__((ext_vector_type(8))) short s16x8v;

template <int OUTF32, int EPI, int MT, int NTL>
__device__ __forceinline__ void store_tile(f32x4 (&acc)[MT][NTL], const GemmArgs& p, const TileInfo& T, char* scratch,
                                           int wm, int wn, int lane_in) {
  static_assert(NTL == 4, "wave tiles are 64 columns wide");
  // opaque copy: keeps every lane-derived address of the epilogue INSIDE the persistent tile loop; hoisted out
  // of it (they are tile-invariant) they would sit in VGPRs across the main loop and push it into scratch
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
  const int g = lane >> 4, i = lane & 15;
  if (OUTF32) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int m = T.m0 + wm * MT * 16 + mi * 16 + i;
      if (m >= p.M) continue;
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni) {
        const int n = T.n0 + wn * NTL * 16 + ni * 16 + 4 * g;
        if (n >= p.N) continue;
        f32x4 v = acc[mi][ni];
        if (p.bias != nullptr && T.slice == 0) v += *(const f32x4*)(p.bias + n);
        float* dst = p.splitk > 1 ? p.ws + ((long)T.slice * p.M + m) * p.N + n : (float*)p.C + (long)m * p.ldc + n;
        if (p.splitk == 1 && p.accumulate) v += *(const f32x4*)dst;
        *(f32x4*)dst = v;
      }
    }
  } else {
    const int mb = T.m0 + wm * MT * 16, nb = T.n0 + wn * 64;
    const int rows_valid = p.M - mb, cols_valid = p.N - nb;          // may be <= 0 or beyond the wave tile
    const bool full = rows_valid >= MT * 16 && cols_valid >= 64;     // wave-uniform
    const int rr = lane >> 3, cc = lane & 7;                         // 16-byte line view: row rr (+8), chunk cc
    char* cbase = (char*)((bf16_t*)p.C + (long)mb * p.ldc + nb);     // wave-uniform
    const unsigned line_c = (unsigned)((rr * p.ldc + cc * 8) * 2);

    f32x4 b4[NTL];
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni) {
      const int n = nb + ni * 16 + 4 * g;
      b4[ni] = (p.bias != nullptr && n < p.N) ? *(const f32x4*)(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // aux operands are fetched in chunks of CM row-blocks, one chunk ahead of the rows being stored, so a chunk's
    // loads are always issued before the previous chunk's stores and at most two chunks sit in registers
    constexpr int CM = MT >= 8 ? 2 : MT;
    constexpr int NCH = MT / CM;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI >= 2 ? p.aux + (long)mb * p.ldaux + nb : (const bf16_t*)p.C), 0, (int)OOB_OFF, 0x00020000);
    const unsigned frag_x = (unsigned)((i * p.ldaux + 4 * g) * 2);                                   // EPI 3: fragment view
    const unsigned line_x = (cc * 8 < cols_valid) ? (unsigned)((rr * p.ldaux + cc * 8) * 2) : OOB_OFF;   // EPI 2: line view
    u32x2 ax[2][EPI == 3 ? CM : 1][NTL];
    u32x4 al[2][EPI == 2 ? CM : 1][2];
    auto load_chunk = [&](int c, int buf) {
#pragma unroll
      for (int q = 0; q < CM; ++q) {
        const int mi = c * CM + q;
        if (EPI == 3) {       // + aux (residual), added in fp32 before the single rounding
#pragma unroll
          for (int ni = 0; ni < NTL; ++ni) {
            const bool ok = full || (mi * 16 + i < rows_valid && ni * 16 + 4 * g < cols_valid);
            ax[buf][q][ni] = __builtin_amdgcn_raw_buffer_load_b64(rx, ok ? (int)frag_x : (int)OOB_OFF, (int)((mi * 16 * p.ldaux + ni * 16) * 2), 0);
          }
        }
        if (EPI == 2) {       // ReLU mask (aux > 0): commutes with the bf16 rounding, applied on the 16-byte lines
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const bool ok = full || mi * 16 + k * 8 + rr < rows_valid;
            al[buf][q][k] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (int)line_x : (int)OOB_OFF, (int)((mi * 16 + k * 8) * p.ldaux * 2), 0);
          }
        }
      }
    };
    // per-lane LDS addresses: fragment writes (chunk (2ni + g/2) ^ (i&7)) and line reads (chunk cc ^ (row&7))
    char* wr = scratch + i * 128 + (g & 1) * 8;
    const int wsw = i & 7, gh = g >> 1;
    const char* rd0 = scratch + rr * 128 + ((cc ^ rr) << 4);          // rows rr and rr+8 share row&7
    if (EPI >= 2) load_chunk(0, 0);
    float cs8[EPI == 2 ? 8 : 1];          // EPI 2: running column sums of this lane's 16-byte line position (8 columns), fp32
#pragma unroll
    for (int e = 0; e < (EPI == 2 ? 8 : 1); ++e) cs8[e] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (EPI >= 2 && c + 1 < NCH) load_chunk(c + 1, (c + 1) & 1);
#pragma unroll
      for (int q = 0; q < CM; ++q) {
        const int mi = c * CM + q;
#pragma unroll
        for (int ni = 0; ni < NTL; ++ni) {
          f32x4 v = acc[mi][ni] + b4[ni];
          if (EPI == 3) {
            const bf16x4 a4 = __builtin_bit_cast(bf16x4, ax[c & 1][q][ni]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)a4[e];
          }
          bf16x4 o = __builtin_convertvector(v, bf16x4);
          if (EPI == 1) {       // ReLU on the rounded values: max as int16 clears every negative (and -0)
            const s16x4v z = {0, 0, 0, 0};
            o = __builtin_bit_cast(bf16x4, __builtin_elementwise_max(__builtin_bit_cast(s16x4v, o), z));
          }
          *(LDS_PTR(bf16x4))(wr + (((ni * 2 + gh) ^ wsw) << 4)) = o;
        }
        // LDS stores hand their data over on a separate path and a later read of the same wave may overtake
        // them: retire the stores first.  (The reads are retired before the next row-block's stores anyway:
        // the global stores below consume them.)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        u32x4 ln[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) ln[k] = *(LDS_PTR(const u32x4))(rd0 + k * 1024);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          u32x4 v = ln[k];
          if (EPI == 2) {
            const s16x8v z = {0, 0, 0, 0, 0, 0, 0, 0};
            const s16x8v neg = __builtin_elementwise_sub_sat(z, __builtin_bit_cast(s16x8v, al[c & 1][q][k]));   // sign set <=> aux > 0
            v &= __builtin_bit_cast(u32x4, neg >> 15);
            // pin the masked line HERE: sunk into the predicated store below, the aux load would stay unretired on
            // the not-taken path and hipcc would guard every fragment register of the K loop with s_waitcnt vmcnt
            asm volatile("" : "+v"(v));
            if (p.cs != nullptr) {           // rows / columns outside the matrix were masked to zero by the range-checked aux load
              const bf16x8 m8 = __builtin_bit_cast(bf16x8, v);
#pragma unroll
              for (int e = 0; e < 8; ++e) cs8[e] += (float)m8[e];
            }
          }
          const bool ok = mi * 16 + k * 8 + rr < rows_valid && cc * 8 < cols_valid;
          // plain global store, uniform base + per-lane 32-bit offset.  (A raw-buffer store with the same
          // offsets loses lanes on this path, nondeterministically, on 8-wave blocks; measured, not understood.)
          if (full || ok) *(u32x4*)(cbase + (long)((mi * 16 + k * 8) * p.ldc * 2) + line_c) = v;
          if (EPI <= 1 && p.CQ != nullptr) {       // kernel-uniform: MX-fp8 copy of the stored (rounded) values for the NEXT GEMM
            // a 32-column MX block = the 16-byte lines of lanes cc, cc^1, cc^2, cc^3 of one row: mx8_block's two xor-shuffles
            float f8[8];
            bf8_to_f32(__builtin_bit_cast(bf16x8, v), f8);
            int e;
            const __attribute__((ext_vector_type(2))) int w8 = mx8_block(f8, e);       // every lane takes part
            const long qrow = mb + mi * 16 + k * 8 + rr;
            const int qcol = nb + cc * 8;
            if (full || ok) {
              *(__attribute__((ext_vector_type(2))) int*)(p.CQ + qrow * p.ldcq + qcol) = w8;
              if ((cc & 3) == 0) p.SC[(long)(qcol >> 5) * p.ldsc + qrow] = (uint8_t)e;
            }
          }
        }
      }
    }
    if (EPI == 2 && p.cs != nullptr) {
      // lanes rr = 0..7 hold the same 8 columns (cc): fold them (lane = rr * 8 + cc -> xor 8, 16, 32), lane rr == 0 stores
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = cs8[e];
        t += __shfl_xor(t, 8);
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        cs8[e] = t;
      }
      if (rr == 0 && mb < p.M && cc * 8 < cols_valid) {
        float* row = p.cs + (long)(mb / (MT * 16)) * p.N + nb + cc * 8;
        *(f32x4*)row = (f32x4){cs8[0], cs8[1], cs8[2], cs8[3]};
        *(f32x4*)(row + 4) = (f32x4){cs8[4], cs8[5], cs8[6], cs8[7]};
      }
    }
  }
}
