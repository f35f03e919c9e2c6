"""Inference (eval, no_grad) forward of FusionWithEmotionDecoder at cfg 2 (d=768, T_a=400, T_t=128, N_e=6, B=64): the bf16 product
path against the fp32-tolerance mode (HRIEMO_PRECISION=fp32: three bf16 products per fp32 product on the GEMM kernel, fp32 MFMA
attention cores), eager launches, median of 10."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd as H
torch.manual_seed(1234)
B, Ta, Tt, d = 64, 400, 128, 768
m = H.FusionWithEmotionDecoder(d_model=d, num_emotions=6, n_heads=8, dropout=0.1).cuda().eval()
g = torch.Generator().manual_seed(1)
h_a, h_t = torch.randn(B, Ta, d, generator=g).cuda(), torch.randn(B, Tt, d, generator=g).cuda()
outs = {}
for mode in ("bf16", "fp32"):
    H.set_precision(mode)
    with torch.no_grad():
        for _ in range(3): out = m(h_a, h_t)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); out = m(h_a, h_t); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    outs[mode] = [o.float() for o in out]
    ts.sort()
    fl = 65.378e9 / 3 * B       # forward third of SURVEY 8(d)'s fwd+bwd FLOPs per utterance
    print(f"{mode}: eval forward {ts[len(ts) // 2]:.3f} ms (B={B}) = {B / ts[len(ts) // 2] * 1e3:.0f} utt/s, {fl / ts[len(ts) // 2] / 1e9:.0f} model TFLOP/s", flush=True)
for name, a, b in zip(("logits", "beta", "z"), outs["bf16"], outs["fp32"]):
    print(f"  bf16 path vs fp32 mode, {name}: max abs diff {float((a - b).abs().max()):.2e} (max |fp32| {float(b.abs().max()):.2f})")
