#!/usr/bin/env python3
"""`make check-isa`: audit of the device ISA of every code object of libhriemo.so.

1. FAILS on the instruction form behind round 2's wrong dS in the single-pass attention backward: a packed fp32
   VALU instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) whose LOW result selects the HIGH dword of src1
   (`op_sel:[x,1...]`).  On MI355X that low-half result sporadically came back as if src1 were 0 in lanes 48..63 at two waves per
   SIMD (scripts_dev/forensics: the fault follows the half, not the data; s_nop on either side, a drained LDS queue and a
   VALU-written source pair do not change it; the same subtraction with the cross select on src0, without a cross select, as
   v_pk_fma_f32 with the select on src2, or as two v_sub_f32 is exact).  hipcc's SLP vectoriser produces the form from two
   scalar subtractions that share their subtrahend, so attention.hip is compiled with -fno-slp-vectorize.
2. Prints, per kernel, registers / LDS / scratch and the spill counts, and how many scratch instructions sit inside loops;
   FAILS when a kernel spills inside a loop unless it is listed in LOOP_SPILL_OK (opt-in tuning variants, not default paths).
usage: check_isa.py <name>_dev.s ..."""
import re, sys

# kernels that are known to spill inside a loop (a speed problem, never a correctness one: scratch traffic is counted by vmcnt
# like any other memory operation); none of them is on the default path of BASELINE configs[1]
LOOP_SPILL_OK = (
    "attn_bwd_dkv_kernelILi96ELi4ELi2ELi32ELb1ELb1ELb1E",   # HRIEMO_ATTN_PAIR=1 (512-thread pairing, opt-in, measured slower)
    "attn_bwd_dkv_kernelILi96ELi4ELi2ELi32ELb0ELb1ELb1E",
    "gemm_mx8_kernel",                                       # HRIEMO_GEMM=mx_fp8 (opt-in)
    "ln_pool_bwd_kernelILi8E",                               # d_model > 2048 (no BASELINE config)
)
PK = re.compile(r"^\s*(v_pk_(?:add|mul|fma)_f32)\b.*?\bop_sel:\[([01](?:,[01])+)\]")
bad = 0
for path in sys.argv[1:]:
    text = open(path).read()
    lines = text.split("\n")
    kernel, depth = None, 0
    in_loop_scratch, cross = {}, {}
    for ln in lines:
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", ln)
        if m and not ln.startswith(".L"):
            kernel, depth = m.group(1), 0
            continue
        if ln.startswith(".LBB"):
            d = re.search(r"Depth=(\d+)", ln)
            depth = int(d.group(1)) if d else 0
            continue
        if kernel is None:
            continue
        if "scratch_" in ln and depth > 0:
            in_loop_scratch[kernel] = in_loop_scratch.get(kernel, 0) + 1
        p = PK.match(ln)
        if p:
            sel = p.group(2).split(",")
            cross[kernel] = cross.get(kernel, 0) + 1
            if sel[1] == "1":
                print(f"{path}: {kernel}: forbidden packed fp32 form (low result <- high dword of src1): {ln.strip()}")
                bad += 1
    meta = text.split("amdhsa.kernels:")[-1] if "amdhsa.kernels:" in text else ""
    rows = []
    for blk in meta.split("  - .agpr_count:")[1:]:
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        rows.append((name, g("vgpr_count"), int(blk.split()[0]), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"),
                     g("vgpr_spill_count"), g("sgpr_spill_count"), in_loop_scratch.get(name, 0)))
    spilling = [r for r in rows if r[6] or r[5] or r[8]]
    print(f"{path}: {len(rows)} kernels, {sum(cross.values())} packed fp32 instructions with a low-half operand select (src0 / src2 only), "
          f"{len(spilling)} kernels with scratch")
    for r in spilling:
        ok = r[8] == 0 or any(t in r[0] for t in LOOP_SPILL_OK)
        print(f"   {('ok  ' if r[8] == 0 else 'WARN') if ok else 'FAIL'} {r[0]}: vgpr {r[1]} agpr {r[2]} sgpr {r[3]} lds {r[4]} scratch {r[5]} B, spilled vgpr {r[6]} sgpr {r[7]}, "
              f"scratch instructions inside loops: {r[8]}")
        bad += not ok
sys.exit(1 if bad else 0)
