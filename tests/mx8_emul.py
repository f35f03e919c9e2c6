"""Host emulation of the MX-fp8 quantiser of hri-emo_amd/csrc/gemm_mx8.hip (test infrastructure): OCP microscaling with
block size 32 along the last dimension, e4m3fn elements, E8M0 scales chosen as 2^ceil(log2(amax/448)) so that no element
saturates, round-to-nearest-even conversion (torch.float8_e4m3fn)."""
import torch


def mx8_scale_exponent(amax):
    """unbiased exponent e of the block scale 2^e for block maxima `amax` (fp32 tensor); amax == 0 -> -127"""
    m, ex = torch.frexp(amax.float())                 # amax = m * 2^ex, m in [0.5, 1)
    # amax <= 1.75 * 2^(e+8)  <=>  2m <= 1.75 ? e = ex-1-8 : e = ex-8
    e = torch.where(m * 2 <= 1.75, ex - 9, ex - 8)
    e = torch.where(amax == 0, torch.full_like(e, -127), e.clamp(-126, 126))
    return e.to(torch.int32)


def mx8_quantize(x):
    """x [..., K] (K % 32 == 0) -> (bytes uint8 [..., K], biased scale bytes uint8 [..., K/32])"""
    x = x.float()
    K = x.shape[-1]
    xb = x.reshape(*x.shape[:-1], K // 32, 32)
    amax = xb.abs().amax(-1)
    e = mx8_scale_exponent(amax)
    inv = torch.where(amax == 0, torch.zeros_like(amax), torch.exp2(-e.float()))
    q = (xb * inv[..., None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).reshape(x.shape), (e + 127).to(torch.uint8)


def mx8_dequantize(q, sb):
    K = q.shape[-1]
    v = q.view(torch.float8_e4m3fn).float().reshape(*q.shape[:-1], K // 32, 32)
    return (v * torch.exp2(sb.float() - 127.0)[..., None]).reshape(q.shape)


def mx8_roundtrip(x):
    """quantise + dequantise (the operand the scaled MFMA actually multiplies); non-multiples of 32 pass through"""
    if x.shape[-1] % 32 != 0:
        return x
    q, s = mx8_quantize(x)
    return mx8_dequantize(q, s).to(x.dtype)
