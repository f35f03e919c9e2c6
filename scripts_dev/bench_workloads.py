"""Secondary workloads of BASELINE.json (not the headline; bench.py measures cfg 2): fwd+bwd step time on one MI355X.
cfg 4: MOSEI shape d=768, T_a=1000, T_t=50, N_e=6, B=32;  cfg 5: d=1024, 4 fusion + 2 decoder layers, N_e=7, B=32 per GPU
(global 256 over 8 GPUs), once on the bf16 GEMMs and once ("cfg5_fp8") with the forward projection / FFN GEMMs on MX-fp8
operands (HRIEMO_GEMM=mx_fp8 / hri_emo_amd.set_gemm_mode): its own roofline line (dominant fp8 kernel class against the 5 PF
dense fp8 peak).  FLOPs per utterance from SURVEY.md 8(d)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd as H
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
WORK = {
    "cfg2": (dict(d_model=768, num_emotions=6, n_heads=8, num_layers_fusion=2, num_layers_decoder=2), 400, 128, 64, 65.378e9),
    "cfg4": (dict(d_model=768, num_emotions=6, n_heads=8, num_layers_fusion=2, num_layers_decoder=2), 1000, 50, 32, 136.715e9),
    "cfg5": (dict(d_model=1024, num_emotions=7, n_heads=8, num_layers_fusion=4, num_layers_decoder=2), 400, 128, 32, 227.115e9),
    "cfg5_fp8": (dict(d_model=1024, num_emotions=7, n_heads=8, num_layers_fusion=4, num_layers_decoder=2), 400, 128, 32, 227.115e9),
}
for name in sys.argv[1:] or ["cfg4", "cfg5"]:
    cfg, Ta, Tt, B, fl = WORK[name]
    H.set_gemm_mode("mx_fp8" if name.endswith("_fp8") else "bf16")
    torch.manual_seed(1234)
    m = H.FusionWithEmotionDecoder(dropout=0.1, beta_hidden=256, **cfg).cuda().train()
    g = torch.Generator().manual_seed(1234)
    batch = (torch.randn(B, Ta, cfg["d_model"], generator=g).cuda().bfloat16(), torch.randn(B, Tt, cfg["d_model"], generator=g).cuda().bfloat16(),
             torch.zeros(B, Ta, dtype=torch.bool, device="cuda"), torch.zeros(B, Tt, dtype=torch.bool, device="cuda"),
             (torch.rand(B, cfg["num_emotions"], generator=g) < 0.3).float().cuda())
    dp = DataParallelStep(m, fusion_step_loss, overlap=False)
    dp.set_global_batch(B)
    dp.step(*batch); dp.capture(*batch)
    for _ in range(5): dp.step(*batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dp.step(*batch)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{name}: d={cfg['d_model']} T_a={Ta} T_t={Tt} B={B}: {ms:.3f} ms/step = {B / ms * 1e3:.0f} utt/s = {B / ms * 1e3 * fl / 1e12:.0f} model TFLOP/s "
          f"({B / ms * 1e3 * fl / 1e12 / 25:.1f} % of bf16 MFMA peak)", flush=True)
    if name.endswith("_fp8"):
        # roofline of the fp8 kernel class: HIP events around each launch (libhriemo prof hooks), eager, one stream
        import ctypes, json
        from hri_emo_amd import _lib, _ops
        L = _lib.lib()
        two, _ops.TWO_STREAMS = _ops.TWO_STREAMS, False
        L.hriemo_prof_enable(1)
        for _ in range(3): dp._fwd_bwd(*batch)
        torch.cuda.synchronize()
        rows = {}
        for c in range(L.hriemo_prof_nclass()):
            msum, n, work = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
            L.hriemo_prof_collect(c, ctypes.byref(msum), ctypes.byref(n), ctypes.byref(work))
            rows[L.hriemo_prof_name(c).decode()] = (msum.value, n.value, work.value)
        L.hriemo_prof_enable(0)
        _ops.TWO_STREAMS = two
        ms8, n8, w8 = rows["gemm_mx8_nt"]
        print(json.dumps({"workload": "cfg5 d=1024 4+2 layers N_e=7 B=32, forward projection/FFN GEMMs on MX-fp8", "ms_per_step": round(ms, 3),
                          "roofline": {"bound": "mfma", "kernel": "gemm_mx8_nt", "achieved": round(w8 / (ms8 * 1e-3) / 1e12, 1), "peak": 5000.0, "unit": "TFLOP/s",
                                       "frac": round(w8 / (ms8 * 1e-3) / 1e12 / 5000.0, 4), "launches_per_step": n8 // 3, "avg_launch_us": round(ms8 / max(n8, 1) * 1e3, 2),
                                       "algorithmic_flop_per_launch": round(w8 / max(n8, 1))},
                          "device_ms_per_step_by_class": {k: round(v[0] / 3, 3) for k, v in rows.items() if v[1]}}), flush=True)
    del dp, m, batch
    torch.cuda.empty_cache()
