"""Forensics of round 2's sporadic dS fault in the single-pass attention backward (csrc/attention.hip, FUSED + BITS, KW = 2).
Generates variants of attention.hip that differ ONLY in how  dif = dP~ - delta'  is computed for the wave's two key sub-tiles,
builds one libhriemo_k<N>.so per variant (into this directory; shared objects travel to the GPU box, they are not tracked) and
leaves the device ISA of every variant next to it.  scripts_dev/forensics/run_variants.py compares them on hardware.
  k0  scalar v_sub_f32 pinned by asm (round 2's guard)               -- control
  k1  plain C subtraction: hipcc SLP-packs the two sub-tiles into
      v_pk_add_f32 d, a, ld op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]   (a = {kw1, kw0}; ld = the (lse', delta') pair as loaded)
  k2  the same instruction written out, a = {kw0, kw1}               -- does the fault follow the half or the sub-tile?
  k3  the same instruction written out, a = {kw1, kw0}               -- the compiler's pairing
  k4  v_pk_add_f32 d, a, dd neg_lo:[0,1] neg_hi:[0,1], dd = {delta', delta'}, a = {kw1, kw0}    -- no cross-half operand select
  k5  k3 behind s_nop 7                                              -- does distance to the producers matter?
  k6  k3 followed by s_nop 7                                         -- does distance to the consumers matter?
  k7  k1 compiled with -fno-slp-vectorize (no packed fp32 anywhere)
  k10 k3 with s_waitcnt lgkmcnt(0) in front (no LDS return of this wave in flight when it executes)
  k11 k3 on a VALU-written copy of the (lse', delta') pair (v_mov x2) instead of the registers the ds_read filled
  k12 cross-half select without negation: v_pk_add_f32 d, a, nld op_sel:[0,1], nld = {lse', -delta'}
  k13 cross-half select on src0: v_pk_add_f32 d, ld, a op_sel:[1,0] neg_lo:[1,0] neg_hi:[1,0]
  k15 v_pk_fma_f32 d, a, one, ld op_sel:[0,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]   (a * 1 - delta' with the cross select on src2)
  k8  the shipped loop nest untouched, only the asm pin replaced by the plain C subtraction (= round 2's failing build: hipcc
      interleaves the four packed subtracts with the exponentials instead of grouping them)
  k9  k8 compiled with -fno-slp-vectorize                            -- the candidate product form
"""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "..", "hri-emo_amd", "csrc")
src = open(os.path.join(CSRC, "attention.hip")).read()

BEGIN = "#pragma unroll\n      for (int kw = 0; kw < KW; ++kw) {\n        const uint32_t key = (uint32_t)(kbase + kw * 16 + i);"
END = "          *(LDS_PTR(bf16x4))(dSt + (wave * KW * 16 + kw * 16 + i) * DSS + (qs * 16 + 4 * g) * 2) = h4;\n        }\n      }\n"
b = src.index(BEGIN, src.index("void attn_bwd_dkv_kernel("))      # the hash kernel further up has the same loop head
e = src.index(END, b) + len(END)
assert src.count(BEGIN) == 2 and src.count(END) == 1

NEW = r'''
      float pk_[KW][4], dpd_[KW][4], dif_[KW][4];
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        const uint32_t key = (uint32_t)(kbase + kw * 16 + i);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pk = kvalid[kw] ? EXP2(fmaf(s[kw][r], sl2, lse4[r])) : 0.f;
          float pd = pk, dpd = dp[kw][r];
          if (BITS || a.thr16 != 0) {
            bool keep;
            if (BITS) {
              keep = ((mk4[r] >> (mbit0 + 4 * kw)) & 1u) != 0u;
            } else {
              const uint32_t x = mix24(hkb[kw] + (uint32_t)(qt * QT + qs * 16 + 4 * g + r) * DROP_CA);
              keep = (key & 1u) ? keep_hi(x, a.thr16) : keep_lo(x, a.thr16);
            }
            pd = keep ? pk : 0.f;
            dpd = keep ? dpd : 0.f;
          }
          pk_[kw][r] = pk; dpd_[kw][r] = dpd;
          pf[kw][qs >> 1][(qs & 1) * 4 + r] = (bf16_t)pd;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        if constexpr (KW == 2 && VARIANT >= 2) {
          const f32x2 ld = {lse4[r], del4[r]};
          f32x2 d;
          if constexpr (VARIANT == 2) {
            const f32x2 av = {dpd_[0][r], dpd_[1][r]};
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(av), "v"(ld));
            dif_[0][r] = d[0]; dif_[1][r] = d[1];
          } else {
            const f32x2 av = {dpd_[1][r], dpd_[0][r]};
            if constexpr (VARIANT == 3) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(av), "v"(ld));
            if constexpr (VARIANT == 4) { const f32x2 dd = {del4[r], del4[r]}; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(av), "v"(dd)); }
            if constexpr (VARIANT == 5) asm("s_nop 7\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(av), "v"(ld));
            if constexpr (VARIANT == 10) asm("s_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(av), "v"(ld));
            if constexpr (VARIANT == 11) { f32x2 lc; asm("v_mov_b32 %0, %1" : "=v"(lc[0]) : "v"(ld[0])); asm("v_mov_b32 %0, %1" : "=v"(lc[1]) : "v"(ld[1])); asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(av), "v"(lc)); }
            if constexpr (VARIANT == 12) { const f32x2 nld = {lse4[r], -del4[r]}; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(av), "v"(nld)); }
            if constexpr (VARIANT == 13) asm("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "=v"(d) : "v"(av), "v"(ld));
            if constexpr (VARIANT == 15) { const f32x2 one = {1.f, 1.f}; asm("v_pk_fma_f32 %0, %1, %3, %2 op_sel:[0,0,1] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(av), "v"(ld), "v"(one)); }
            if constexpr (VARIANT == 6) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 7" : "=v"(d) : "v"(av), "v"(ld));
            dif_[1][r] = d[0]; dif_[0][r] = d[1];
          }
        } else {
#pragma unroll
          for (int kw = 0; kw < KW; ++kw) {
            if constexpr (VARIANT == 0) asm("v_sub_f32 %0, %1, %2" : "=v"(dif_[kw][r]) : "v"(dpd_[kw][r]), "v"(del4[r]));
            else dif_[kw][r] = dpd_[kw][r] - del4[r];
          }
        }
      }
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dsv = pk_[kw][r] * dif_[kw][r];
          if (FUSED) dsum[kw] += dsv;
          dsf[kw][qs >> 1][(qs & 1) * 4 + r] = (bf16_t)dsv;
        }
        if (FUSED) {
          const bf16x8 f = dsf[kw][qs >> 1];
          const bf16x4 h4 = (qs & 1) ? (bf16x4){f[4], f[5], f[6], f[7]} : (bf16x4){f[0], f[1], f[2], f[3]};
          *(LDS_PTR(bf16x4))(dSt + (wave * KW * 16 + kw * 16 + i) * DSS + (qs * 16 + 4 * g) * 2) = h4;
        }
      }
'''

PIN = 'asm("v_sub_f32 %0, %1, %2" : "=v"(dif) : "v"(dpd), "v"(del4[r]));'
assert src.count(PIN) == 1

def variant_source(k):
    if k in (8, 9):
        return src.replace(PIN, "dif = dpd - del4[r];")
    body = NEW.replace("VARIANT", str(k if k != 7 else 1))
    return src[:b] + body + src[e:]

FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1".split()
ks = [int(x) for x in sys.argv[1:]] or list(range(14)) + [15]
objs = [os.path.join(CSRC, "build", f"{n}.o") for n in ("gemm", "gemm_mx8", "rowops", "fp32mode", "runtime")]
for o in objs:
    assert os.path.exists(o), f"{o}: run make -C hri-emo_amd/csrc first"
procs = []
for k in ks:
    d = os.path.join(HERE, f"k{k}")
    os.makedirs(d, exist_ok=True)
    for h in ("common.h", "gemm_common.h"):
        open(os.path.join(d, h), "w").write(open(os.path.join(CSRC, h)).read())
    open(os.path.join(d, "attention.hip"), "w").write(variant_source(k))
    extra = ["-fno-slp-vectorize"] if k in (7, 9) else []
    cmd = (f"cd {d} && /opt/rocm/bin/hipcc {' '.join(FLAGS + extra)} -c attention.hip -o attention.o && "
           f"/opt/rocm/bin/hipcc {' '.join(FLAGS + extra)} --offload-device-only -S attention.hip -o attention_dev.s 2>/dev/null && "
           f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o {HERE}/libhriemo_k{k}.so attention.o {' '.join(objs)}")
    procs.append((k, subprocess.Popen(cmd, shell=True)))
    if len(procs) % 4 == 0:
        for _, p in procs[-4:]: p.wait()
for k, p in procs:
    assert p.wait() == 0, k
    print("built variant", k)
