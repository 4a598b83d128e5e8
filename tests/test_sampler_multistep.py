"""-m "not gpu": runner.inference beyond the pipeline's forced steps = 1 / cfg = 1 (round 6): the Euler sampler over four trailing
timesteps with classifier-free guidance (scale 2.5 on the first half of the steps, rescale 0.7) and a batch of TWO clips of different
sizes, against tests/golden/sampler_multistep.pt = the reference's own EulerSampler / schedule / timesteps / CFG dispatcher wired as
VideoDiffusionInfer.inference wires them (infer.py:315-395) over its NaDiT, each clip as a batch of one (oracle/make_golden.py --only
r6-sampler; the reference's own batched call hands the clips' text to the windows round-robin -- na.py:381-387 -- and is not a mode to
reproduce: see the note there).  Here on the CPU double of the C ABI: exact arithmetic (fp32 storage) pins the host logic,
the bf16 regime bounds what the product's storage adds over four model calls; the same case runs over HipOps in
tests/test_gpu_parity.py.  Also: NaDiTEngine's reference-shaped __call__ with a batch of clips (na.flatten order)."""
import os

import pytest
import torch

from conftest import sub, rel_err, GOLDEN
from ops_reference import TorchOps


def run_case(ops, plain=False):
    from oracle import make_golden as mg
    config, weights, dit, runner = (sub(n) for n in ("config", "weights", "dit", "runner"))
    g = torch.load(os.path.join(GOLDEN, "sampler_multistep.pt"), weights_only=True)
    cfg = config.DIT_TINY
    r = runner.VideoDiffusionInfer(runner.default_config(cfg))
    r.dit = dit.NaDiTEngine(cfg, weights.synth_dit_state_dict(cfg), ops)
    r.config.diffusion.timesteps.sampling.steps = g["steps"]
    r.config.diffusion.cfg.update(scale=1.0 if plain else g["cfg_scale"], partial=g["cfg_partial"], rescale=g["cfg_rescale"])
    r.configure_diffusion()
    assert torch.allclose(r.sampling_timesteps.float(), g["timesteps"].float())
    noises, conds, tp, tn = mg.sampler_inputs()
    dt = ops.act_dtype
    cast = lambda ts: [t.to(dt) for t in ts]
    if plain:
        out = r.inference(cast(noises[:1]), cast(conds[:1]), cast(tp[:1]), cast(tn[:1]))
        return [rel_err(out[0].float().cpu(), g["plain"])]
    outs = r.inference(cast(noises), cast(conds), cast(tp), cast(tn))
    assert [tuple(o.shape) for o in outs] == [tuple(o.shape) for o in g["outs"]]
    return [rel_err(o.float().cpu(), w) for o, w in zip(outs, g["outs"])]


def test_multistep_cfg_batch_exact_arithmetic():
    errs = run_case(TorchOps("cpu", act_dtype=torch.float32)) + run_case(TorchOps("cpu", act_dtype=torch.float32), plain=True)
    print("multi-step Euler + CFG (fp32 storage) vs the reference's sampler:", ["%.2e" % e for e in errs])
    assert max(errs) < 2e-5


def test_multistep_cfg_batch_product_storage():
    errs = run_case(TorchOps("cpu", act_dtype=torch.bfloat16))
    print("multi-step Euler + CFG (bf16 operands, h16 stream) vs the reference's sampler:", ["%.2e" % e for e in errs])
    assert max(errs) < 1.5e-2


def test_engine_call_with_a_batch_of_clips_equals_per_clip_calls():
    """NaDiT.forward's signature with vid_shape (B, 3): rows of the clips concatenated in order (na.flatten), one timestep per clip."""
    from oracle import make_golden as mg
    config, weights, dit = sub("config"), sub("weights"), sub("dit")
    cfg = config.DIT_TINY
    eng = dit.NaDiTEngine(cfg, weights.synth_dit_state_dict(cfg), TorchOps("cpu", act_dtype=torch.float32))
    noises, conds, tp, _ = mg.sampler_inputs()
    vids = [torch.cat([n, c], dim=-1).float() for n, c in zip(noises, conds)]
    shapes = torch.tensor([list(v.shape[:3]) for v in vids])
    flat = torch.cat([v.reshape(-1, v.shape[-1]) for v in vids])
    txt = torch.cat([t.float() for t in tp])
    got = eng(flat, txt, shapes, torch.tensor([[58], [58]]), torch.tensor([750.0, 250.0])).vid_sample
    want = torch.cat([eng.forward(v, t.float(), ts).reshape(-1, 16) for v, t, ts in zip(vids, tp, (750.0, 250.0))])
    assert got.shape == want.shape and torch.equal(got, want)
    with pytest.raises(ValueError):
        eng(flat, txt, shapes, torch.tensor([[58], [57]]), 1000.0)
