"""Secondary workloads of BASELINE.json (not the headline; bench.py measures cfg 2): fwd+bwd step time on one MI355X.
cfg 4: MOSEI shape d=768, T_a=1000, T_t=50, N_e=6, B=32;  cfg 5: d=1024, 4 fusion + 2 decoder layers, N_e=7, B=32
(bf16 here -- the fp8 GEMM path of cfg 5 is not built).  FLOPs per utterance from SURVEY.md 8(d)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd as H
from hri_emo_amd.dp import DataParallelStep
from hri_emo_amd.train import fusion_step_loss
WORK = {
    "cfg2": (dict(d_model=768, num_emotions=6, n_heads=8, num_layers_fusion=2, num_layers_decoder=2), 400, 128, 64, 65.378e9),
    "cfg4": (dict(d_model=768, num_emotions=6, n_heads=8, num_layers_fusion=2, num_layers_decoder=2), 1000, 50, 32, 136.715e9),
    "cfg5": (dict(d_model=1024, num_emotions=7, n_heads=8, num_layers_fusion=4, num_layers_decoder=2), 400, 128, 32, 227.115e9),
}
for name in sys.argv[1:] or ["cfg4", "cfg5"]:
    cfg, Ta, Tt, B, fl = WORK[name]
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
    del dp, m, batch
    torch.cuda.empty_cache()
