"""Closed-form checks of the restated third-party arithmetic (oracle/third_party_shims.py)."""
import math

import torch

from oracle import third_party_shims as tps
from oracle import dit_oracle


def test_lang_freqs_formula():
    r = tps.RotaryEmbedding(dim=42, freqs_for="lang", theta=10000)
    want = torch.tensor([10000.0 ** (-(2 * i) / 42) for i in range(21)])
    assert torch.allclose(r.freqs, want, rtol=1e-6)


def test_axial_freqs_layout_and_rotation_preserves_pair_norms():
    r = tps.RotaryEmbedding(dim=42, freqs_for="lang", theta=10000)
    f = r.get_axial_freqs(5, 4, 3)
    assert f.shape == (5, 4, 3, 126)
    # axis a, frequency i sits at [42a + 2i, 42a + 2i + 1] and equals pos_a * freq_i
    assert torch.allclose(f[3, 2, 1, 42 + 2 * 7], 2 * r.freqs[7])
    assert torch.allclose(f[3, 2, 1, 84 + 2 * 20 + 1], 1 * r.freqs[20])
    x = torch.randn(5 * 4 * 3, 128)
    y = tps.apply_rotary_emb(f.reshape(-1, 126), x)
    assert torch.allclose(y[:, 126:], x[:, 126:])
    n_in = x[:, :126].reshape(-1, 63, 2).norm(dim=-1)
    n_out = y[:, :126].reshape(-1, 63, 2).norm(dim=-1)
    assert torch.allclose(n_in, n_out, atol=1e-5)


def test_oracle_rope_equals_shim_rope():
    r = tps.RotaryEmbedding(dim=42, freqs_for="lang", theta=10000)
    f = r.get_axial_freqs(6, 5, 4)[2:5, :3, :2].reshape(-1, 126)
    pos = torch.stack(torch.meshgrid(torch.arange(2, 5), torch.arange(3), torch.arange(2), indexing="ij"), -1).reshape(-1, 3)
    x = torch.randn(pos.shape[0], 2, 128)
    want = tps.apply_rotary_emb(f, x.transpose(0, 1)).transpose(0, 1)
    got = dit_oracle.apply_rope(x, dit_oracle.rope_angles(r.freqs, pos))
    assert torch.allclose(got, want, atol=1e-6)


def test_timestep_embedding_formula():
    e = tps.get_timestep_embedding(torch.tensor([1000.0]), 256, flip_sin_to_cos=False, downscale_freq_shift=0)
    k = 17
    ang = 1000.0 * math.exp(-math.log(10000.0) * k / 128)
    assert abs(float(e[0, k]) - math.sin(ang)) < 1e-4 and abs(float(e[0, 128 + k]) - math.cos(ang)) < 1e-4
    assert torch.allclose(dit_oracle.timestep_embedding(torch.tensor([1000.0])), e)


def test_attention_shim_vs_explicit_softmax():
    torch.manual_seed(0)
    a = tps.Attention(64, heads=1, dim_head=64, norm_num_groups=32, eps=1e-6, residual_connection=True, bias=True)
    x = torch.randn(2, 64, 5, 7)
    y = a(x)
    h = a.group_norm(x.flatten(2)).transpose(1, 2)
    q, k, v = a.to_q(h), a.to_k(h), a.to_v(h)
    p = torch.softmax(q @ k.transpose(1, 2) / 8.0, dim=-1)
    want = a.to_out[0](p @ v).transpose(1, 2).reshape(2, 64, 5, 7) + x
    assert torch.allclose(y, want, atol=1e-5)
