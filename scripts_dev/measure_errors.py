import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hri_emo_amd as H
from oracle import hri_emo_oracle as O
for (B, Ta, Tt, d, ne, lf, ld) in [(4, 32, 16, 128, 4, 2, 2), (3, 100, 40, 768, 6, 2, 2), (2, 400, 128, 768, 6, 2, 2), (2, 1000, 50, 768, 6, 2, 2), (2, 400, 128, 1024, 7, 4, 2)]:
    torch.manual_seed(1234)
    kw = dict(d_model=d, num_emotions=ne, n_heads=8, dropout=0.1, num_layers_fusion=lf, num_layers_decoder=ld)
    ref = O.FusionWithEmotionDecoder(**kw).eval(); m = H.FusionWithEmotionDecoder(**kw); m.load_state_dict(ref.state_dict()); m.cuda().eval()
    g = torch.Generator().manual_seed(5)
    h_a, h_t = torch.randn(B, Ta, d, generator=g), torch.randn(B, Tt, d, generator=g)
    la = torch.randint(Ta // 2, Ta + 1, (B,), generator=g); lt = torch.randint(Tt // 2, Tt + 1, (B,), generator=g)
    m_a, m_t = torch.arange(Ta)[None] >= la[:, None], torch.arange(Tt)[None] >= lt[:, None]
    with torch.no_grad():
        lr, br, zr = ref(h_a, h_t, m_a, m_t); lg, bg, zg = m(h_a.cuda(), h_t.cuda(), m_a.cuda(), m_t.cuda())
    e = lambda a, b: ((a.cpu().float() - b).abs().max() / max(1.0, b.abs().max().item())).item()
    print(f"d={d} Ta={Ta} Tt={Tt} L={lf}+{ld}: max|err|/max(1,|ref|): logits {e(lg, lr):.2e} beta {e(bg, br):.2e} z {e(zg, zr):.2e}")
