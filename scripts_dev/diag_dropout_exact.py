"""Diagnostic: per-parameter gradient errors of the HIP train step vs the fp32 oracle, dropout 0 and dropout 0.1 with the
masks replayed (see tests/test_gpu_parity.py::test_train_step_with_dropout_equals_the_oracle_under_the_same_masks)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import hashrng
import hri_emo_amd as H
from hri_emo_amd import _ops
from oracle import hri_emo_oracle as O
import test_gpu_parity as T

B, Ta, Tt, d, ne = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (3, 100, 40, 256, 5))]
for p in (0.0, 0.1):
    torch.manual_seed(1234)
    kw = dict(d_model=d, num_emotions=ne, n_heads=8, dropout=p)
    ref = O.FusionWithEmotionDecoder(**kw).train()
    m = H.FusionWithEmotionDecoder(**kw); m.load_state_dict(ref.state_dict()); m.cuda().train()
    h_a, h_t, m_a, m_t = T._rand_batch(B, Ta, Tt, d, 11)
    y = (torch.rand(B, ne, generator=torch.Generator().manual_seed(12)) < 0.3).float()
    word = int(_ops.seed_word(torch.device("cuda", 0)).item()) & ((1 << 64) - 1)
    log = []; _ops.DROP_LOG = log
    torch.manual_seed(77)
    out_m = T._train_step(m, T.cu(h_a), T.cu(h_t), T.cu(m_a), T.cu(m_t), T.cu(y))
    _ops.DROP_LOG = None
    cursor = [0]
    def replay(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0: return x
        e = log[cursor[0]]; cursor[0] += 1
        seed = (e[1] + word) & ((1 << 64) - 1)
        k = hashrng.attn_mask(seed, e[2], *e[3:7], e[7], e[8]) if e[0] == "attn" else hashrng.rows_mask(seed, e[2], e[3], e[4], e[5], e[6])
        pp = e[7] if e[0] == "attn" else e[5]
        return x * (torch.from_numpy(k.reshape(tuple(x.shape))).to(x.dtype) * hashrng.inv_keep(pp))
    orig = torch.nn.functional.dropout
    torch.nn.functional.dropout = replay
    out_r = T._train_step(ref, h_a, h_t, m_a, m_t, y)
    cursor[0] = 0
    out_y = T._train_step(ref, h_a, h_t, m_a, m_t, y, autocast_cpu=True)
    torch.nn.functional.dropout = orig
    gm, gr, gy = out_m[4], out_r[4], out_y[4]
    rows = sorted(((T._rel(gm[n], gr[n]), T._rel(gy[n], gr[n]), n) for n in gr), reverse=True)
    print(f"p={p}: loss mine {float(out_m[0]):.6f} oracle {float(out_r[0]):.6f}; worst parameters (mine, yardstick, name):")
    for r in rows[:8]: print("   %.4f %.4f %s" % r)
