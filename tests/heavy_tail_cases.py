"""Shared body of the heavy-tail robustness tests (round 6): the engines on weights with the statistics a trained checkpoint can hold
and synthetic N(0, 1 / fan_in) weights never show -- outlier channels in the NaDiT's residual stream (x300), modulation scales and
GroupNorm gains over three decades, O(1) GroupNorm biases, one conv with 10x weights, e4m3-valued matrices -- against the REFERENCE's
fp32 output on the same weights and inputs (tests/golden/heavy_tail.pt, oracle/make_golden.py --only r6-heavy), with the reference's
own bf16 run as the yardstick.  Two levels:
  "tail"      everything stays inside h16's range (+-4.2e6): the overflow guards must stay SILENT, engine error <= reference-bf16 error
              (or the stated floor where the reference's bf16 run happens to land below this storage regime's own rounding);
  "overflow"  one producer pushes the residual stream / trunk beyond h16's range while bf16 / fp32 hold it: the guard must notice,
              re-run the call with fp32 stores ON THE SAME BACKEND, warn, count the re-run, and still match.
Called from tests/test_heavy_tail.py (CPU: the torch double of the C ABI) and tests/test_gpu_parity.py (MI355X: HipOps).
normalization.py:88-109, causal_inflation_lib.py:354-409, compatibility.py:895-938."""
import os
import warnings

import torch

from conftest import sub, rel_err, GOLDEN

BF16 = torch.bfloat16


def _golden():
    return torch.load(os.path.join(GOLDEN, "heavy_tail.pt"), weights_only=True)


def dit_case(ops, level, floor=6e-3):
    from oracle import make_golden as mg
    config, weights, dit = sub("config"), sub("weights"), sub("dit")
    g = _golden()
    cfg = config.DIT_TINY
    sd = mg.heavy_dit_state_dict(weights.synth_dit_state_dict(cfg), level)
    txt = torch.load(os.path.join(GOLDEN, "text_pos_emb.pt"), weights_only=True)
    vid = mg.dit_inputs(*g["dit"]["latent"], seed=g["dit"]["seed_input"])
    eng = dit.NaDiTEngine(cfg, sd, ops)
    assert eng.hid_store == "h16"
    dev = ops.device
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        out = eng.forward(vid.to(dev), txt.to(dev), 1000.0).float().cpu()
    e = rel_err(out, g[f"dit_{level}"])
    e_ref = rel_err(g[f"dit_{level}_refbf16"].float(), g[f"dit_{level}"])
    fired = [w for w in caught if "h16 residual stream" in str(w.message)]
    print(f"NaDiT heavy-tail [{level}] on {ops.name if hasattr(ops, 'name') else type(ops).__name__}: rel-err {e:.3e} "
          f"(reference bf16: {e_ref:.3e}), guard re-runs {eng.overflow_reruns}")
    assert torch.isfinite(out).all()
    assert e <= max(e_ref, floor), (e, e_ref)
    if level == "tail":
        assert eng.overflow_reruns == 0 and not fired
    else:
        assert eng.overflow_reruns == 1 and len(fired) == 1 and eng.hid_store == "h16"
    return e, e_ref


def vae_case(ops, level, floor_enc=1.2e-2, floor_dec=1.2e-2):
    from oracle import make_golden as mg
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    g = _golden()
    cfg = config.VAE_V3
    sd = mg.heavy_vae_state_dict(weights.synth_vae_state_dict(cfg), level)
    eng = vae.VideoVAEEngine(cfg, sd, ops)
    assert "h16" in (eng.trunk_store, eng.branch_store)
    hv, dev = g["vae"], ops.device
    x = mg.blocky_frames(*hv["frames"], seed=hv["seed_x"], cell=hv["cell"])[0].to(dev)
    z = (mg.latent_input(*hv["latent"], seed=hv["seed_z"])[0].permute(1, 2, 3, 0).float() * cfg.scaling_factor).to(BF16).to(dev)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        lat = eng.encode(x).float().cpu()
        runs_enc = eng.overflow_reruns
        y = eng.decode(z).float().cpu()
    want_enc = g[f"vae_{level}_enc"][0].permute(1, 2, 3, 0) * cfg.scaling_factor
    ref_enc = g[f"vae_{level}_enc_refbf16"][0].float().permute(1, 2, 3, 0) * cfg.scaling_factor
    want_dec, ref_dec = g[f"vae_{level}_dec"][0], g[f"vae_{level}_dec_refbf16"][0].float()
    e_enc, e_dec = rel_err(lat, want_enc), rel_err(y, want_dec)
    r_enc, r_dec = rel_err(ref_enc, want_enc), rel_err(ref_dec, want_dec)
    print(f"VAE heavy-tail [{level}]: encode rel-err {e_enc:.3e} (reference bf16 {r_enc:.3e}), decode {e_dec:.3e} (reference bf16 "
          f"{r_dec:.3e}); guard re-runs encode {runs_enc}, decode {eng.overflow_reruns - runs_enc}")
    assert torch.isfinite(lat).all() and torch.isfinite(y).all()
    assert e_enc <= max(r_enc, floor_enc) and e_dec <= max(r_dec, floor_dec), (e_enc, r_enc, e_dec, r_dec)
    if level == "tail":
        assert eng.overflow_reruns == 0 and not caught, [str(w.message)[:80] for w in caught]
    else:
        assert runs_enc == 1 and eng.overflow_reruns == 2 and len(caught) >= 2
    return e_enc, e_dec
