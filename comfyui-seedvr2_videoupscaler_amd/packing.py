"""Checkpoint tensors (reference layouts) -> the layouts the HIP kernels consume.

* nn.Linear weight [N, K]          -> bf16 [ceil128(N), ceil64(K)], zero padded (rows are K-contiguous,
                                       exactly what the MFMA B operand wants; no transpose needed)
* SwiGLU proj_in_gate / proj_in     -> one [2*hidden, K] matrix, rows interleaved in blocks of 16
                                       (gate16 | in16) so the GEMM epilogue sees both halves in one lane
* Conv3d weight [Co, Ci, kt, kh, kw] -> [ceil128(Co), taps*Ci] tap-major / channel-minor (implicit-GEMM K axis)
* biases, norm gains                 -> fp32
"""
import torch

BF16 = torch.bfloat16


def _ceil(x, m):
    return (x + m - 1) // m * m


def pack_matrix(w: torch.Tensor, device, k_pad_to: int = 64) -> torch.Tensor:
    n, k = w.shape
    out = torch.zeros(_ceil(n, 128), _ceil(k, k_pad_to), dtype=BF16, device=device)
    out[:n, :k] = w.to(device=device, dtype=BF16)
    return out


def pack_vec(v: torch.Tensor, device) -> torch.Tensor:
    return v.to(device=device, dtype=torch.float32).contiguous()


def pack_swiglu(w_gate: torch.Tensor, w_in: torch.Tensor, device) -> torch.Tensor:
    h, k = w_gate.shape
    assert w_in.shape == (h, k) and h % 16 == 0
    g = w_gate.to(device=device, dtype=BF16).reshape(h // 16, 16, k)
    u = w_in.to(device=device, dtype=BF16).reshape(h // 16, 16, k)
    inter = torch.stack((g, u), dim=1).reshape(2 * h, k)
    return pack_matrix(inter, device)


def pack_conv3d(w: torch.Tensor, device, cin_pad: int = None) -> torch.Tensor:
    co, ci, kt, kh, kw = w.shape
    w = w.to(device=device, dtype=BF16).permute(0, 2, 3, 4, 1)            # [Co, kt, kh, kw, Ci]
    if cin_pad is not None and cin_pad != ci:
        wp = torch.zeros(co, kt, kh, kw, cin_pad, dtype=BF16, device=device)
        wp[..., :ci] = w
        w = wp
    return pack_matrix(w.reshape(co, -1), device)
