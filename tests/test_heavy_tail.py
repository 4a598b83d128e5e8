"""-m "not gpu": the heavy-tail robustness cases (tests/heavy_tail_cases.py) on the CPU double of the C ABI in the product's storage
regime (bf16 operands, h16 stream / trunk): pins the fixture, the host-side guard logic and the h16 emulation; the same bodies run
over HipOps in tests/test_gpu_parity.py."""
import pytest
import torch

from ops_reference import TorchOps
import heavy_tail_cases as ht


@pytest.mark.parametrize("level", ["tail", "overflow"])
def test_dit_heavy_tail_on_the_cpu_double(level):
    ht.dit_case(TorchOps("cpu", act_dtype=torch.bfloat16), level)


@pytest.mark.parametrize("level", ["tail", "overflow"])
def test_vae_heavy_tail_on_the_cpu_double(level):
    ht.vae_case(TorchOps("cpu", act_dtype=torch.bfloat16), level)
