"""Checkpoint ingest (checkpoint.py) on CPU: safetensors round trips in the registry's precisions,
2D->3D inflation semantics, key checks."""
import os
import warnings

import pytest
import torch

from conftest import sub, rel_err
from ops_reference import TorchOps
from oracle import reference_loader as rl


def test_safetensors_fp16_and_fp8_round_trip(tmp_path):
    from safetensors.torch import save_file
    ck, weights, config = sub("checkpoint"), sub("weights"), sub("config")
    cfg = config.DIT_TINY
    sd = weights.synth_dit_state_dict(cfg, seed=11)
    p16, p8 = str(tmp_path / "seedvr2_ema_3b_fp16.safetensors"), str(tmp_path / "seedvr2_ema_3b_fp8_e4m3fn.safetensors")
    save_file({k: v.to(torch.float16).contiguous() for k, v in sd.items()}, p16)
    sd8 = {k: (v.to(torch.float8_e4m3fn) if v.dim() >= 2 else v.to(torch.float16)).contiguous() for k, v in sd.items()}
    save_file({("model.diffusion_model." + k): v for k, v in sd8.items()}, p8)       # ComfyUI-style prefix
    got16 = ck.prepare_dit_state_dict(ck.load_state_dict(p16), cfg)
    assert set(got16) == set(sd) and all(v.dtype == torch.bfloat16 for k, v in got16.items() if not k.endswith("freqs"))
    assert all(torch.equal(got16[k].float(), sd[k].to(torch.float16).to(torch.float32 if k.endswith("freqs") else torch.bfloat16).float())
               for k in sd)
    assert max(rel_err(got16[k].float(), sd[k].float()) for k in sd) < 1e-3            # (fp16 flushes the tiniest weights)
    got8 = ck.prepare_dit_state_dict(ck.load_state_dict(p8), cfg)
    for k, v in sd8.items():
        assert torch.equal(got8[k].float(), (v if k.endswith("freqs") else v.to(torch.bfloat16)).float()), k
        if v.dtype == torch.float8_e4m3fn:
            assert torch.equal(got8[k].float(), v.float()), k                         # fp8 -> bf16 is an exact up-cast
    with pytest.raises(ValueError):
        ck.load_state_dict(str(tmp_path / "x.gguf"))


def test_engine_from_checkpoint_equals_engine_from_state_dict(tmp_path):
    from safetensors.torch import save_file
    ck, weights, config, dit = sub("checkpoint"), sub("weights"), sub("config"), sub("dit")
    cfg = config.DIT_TINY
    sd = weights.synth_dit_state_dict(cfg, seed=3)
    path = str(tmp_path / "tiny.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, path)                        # bf16: lossless
    ops = TorchOps("cpu", act_dtype=torch.float32)
    g = torch.Generator().manual_seed(0)
    vid, txt = torch.randn(2, 8, 12, 33, generator=g), torch.randn(58, 5120, generator=g)
    a = dit.NaDiTEngine(cfg, sd, ops).forward(vid, txt, 1000.0)
    b = dit.NaDiTEngine(cfg, ck.prepare_dit_state_dict(ck.load_state_dict(path), cfg), ops).forward(vid, txt, 1000.0)
    assert torch.equal(a, b)


def test_missing_keys_and_rope_buffers():
    ck, weights, config = sub("checkpoint"), sub("weights"), sub("config")
    cfg = config.DIT_TINY
    sd = weights.synth_dit_state_dict(cfg)
    no_freqs = {k: v for k, v in sd.items() if not k.endswith("rope.rope.freqs")}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = ck.prepare_dit_state_dict(no_freqs, cfg)
    assert len(w) == cfg.num_layers and all(float(out[f"blocks.{i}.attn.rope.rope.freqs"].abs().sum()) == 0 for i in range(cfg.num_layers))
    broken = dict(sd)
    del broken["vid_in.proj.weight"]
    with pytest.raises(KeyError):
        ck.prepare_dit_state_dict(broken, cfg)


def test_vae_inflation_tail_is_a_per_frame_2d_conv():
    """A tail-inflated kernel on the causally padded clip reproduces the 2D conv frame by frame."""
    ck = sub("checkpoint")
    g = torch.Generator().manual_seed(5)
    w2, x = torch.randn(6, 4, 3, 3, generator=g), torch.randn(1, 4, 5, 9, 9, generator=g)
    w3 = ck.inflate_weight(w2, 3, "tail")
    xp = torch.cat([x[:, :, :1]] * 2 + [x], dim=2)                                     # causal head: frame 0 twice
    y3 = torch.nn.functional.conv3d(torch.nn.functional.pad(xp, (1, 1, 1, 1)), w3)
    y2 = torch.stack([torch.nn.functional.conv2d(x[0, :, t][None], w2, padding=1)[0] for t in range(5)], dim=1)[None]
    assert rel_err(y3, y2) < 1e-6
    assert torch.allclose(ck.inflate_weight(w2, 3, "replicate").sum(dim=2), w2, atol=1e-6)
    if rl.available():
        ns = rl._extract("src/models/video_vae_v3/modules/causal_inflation_lib.py", ["inflate_weight"], {"torch": torch})
        for mode in ("tail", "replicate"):
            assert torch.equal(ns["inflate_weight"](w2, torch.empty(6, 4, 3, 3, 3), mode), ck.inflate_weight(w2, 3, mode))


def test_vae_checkpoint_with_2d_weights_inflates_to_engine_shapes(tmp_path):
    from safetensors.torch import save_file
    ck, weights, config = sub("checkpoint"), sub("weights"), sub("config")
    cfg = config.VAE_TINY
    sd = weights.synth_vae_state_dict(cfg, seed=2)
    flat = {k: (v[:, :, -1].contiguous() if v.dim() == 5 and v.shape[2] == 3 and "resnets.0.conv1" in k else v.contiguous())
            for k, v in sd.items()}                                                    # a few layers stored as 2D kernels
    path = str(tmp_path / "ema_vae_fp16.safetensors")
    save_file(flat, path)                                                               # bf16: lossless
    out = ck.prepare_vae_state_dict(ck.load_state_dict(path), cfg)
    for k, v in sd.items():
        assert out[k].shape == v.shape, k
        if v.dim() == 5 and "resnets.0.conv1" in k and v.shape[2] == 3:
            assert torch.equal(out[k][:, :, -1], v[:, :, -1]) and float(out[k][:, :, :-1].abs().sum()) == 0


def test_7b_checkpoint_is_detected_and_builds_the_7b_engine(tmp_path):
    """A SeedVR2-7B-family state dict (dit_7b: biased GELU MLPs, pixel RoPE, separate weights in every block) written as
    safetensors goes through build_engines() into the DIT_7B graph: the family is detected from the tensors, the 10 RoPE
    frequencies per axis (not the 3B's 21) are zero-filled when missing, and the engine equals one built from the dict."""
    from safetensors.torch import save_file
    ck, weights, config, dit = sub("checkpoint"), sub("weights"), sub("config"), sub("dit")
    cfg = config.DIT_7B_TINY
    sd = weights.synth_dit_state_dict(cfg, seed=5)
    assert ck.detect_dit_config(sd) is config.DIT_7B                       # by the mlp biases, whatever the width
    assert ck.detect_dit_config(weights.synth_dit_state_dict(config.DIT_TINY)) is config.DIT_3B
    assert ck.detect_dit_config({}, "seedvr2_ema_7b_fp16.safetensors") is config.DIT_7B
    path = str(tmp_path / "seedvr2_ema_7b_sharp_fp16.safetensors")         # round 1 refused any file name containing "7b"
    save_file({k: v.contiguous() for k, v in sd.items()}, path)
    ops = TorchOps("cpu", act_dtype=torch.float32)
    eng, _ = ck.build_engines(ops, dit_path=path, dit_cfg=cfg)
    g = torch.Generator().manual_seed(0)
    vid, txt = torch.randn(2, 8, 12, 33, generator=g), torch.randn(58, 5120, generator=g)
    assert torch.equal(eng.forward(vid, txt, 1000.0), dit.NaDiTEngine(cfg, sd, ops).forward(vid, txt, 1000.0))
    no_freqs = {k: v for k, v in sd.items() if not k.endswith("rope.rope.freqs")}
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        out = ck.prepare_dit_state_dict(no_freqs, cfg)
    assert out["blocks.0.attn.rope.rope.freqs"].shape == (cfg.rope_freqs,) == (10,)


def test_vae_checkpoint_missing_shortcut_is_reported():
    """The expected key set comes from the real block widths (meta device), so a checkpoint without the 1x1x1
    conv_shortcut of a width-changing resnet is refused at ingest (round 1 derived the keys from an equal-width model and
    the engine then read the residual with the wrong stride); the engine itself refuses too."""
    ck, weights, config, vae = sub("checkpoint"), sub("weights"), sub("config"), sub("vae")
    cfg = config.VAE_TINY                                                   # (64, 64, 128, 128): one width change per side
    sd = weights.synth_vae_state_dict(cfg, seed=2)
    shortcuts = [k for k in sd if "conv_shortcut" in k]
    assert shortcuts and set(ck.vae_expected_keys(cfg)) == set(sd)
    broken = {k: v for k, v in sd.items() if k not in shortcuts[:2]}
    with pytest.raises(KeyError):
        ck.prepare_vae_state_dict(broken, cfg)
    with pytest.raises(KeyError):
        vae.VideoVAEEngine(cfg, broken, TorchOps("cpu", act_dtype=torch.float32))
