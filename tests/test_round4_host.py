"""-m "not gpu", round 4: the h16 storage kind on the host side, the output dtype of the pipeline glue, and the oracle/_ref recipe."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import sub, rel_err, ROOT
from ops_reference import TorchOps, H16, H16_SCALE, _ld, _st
from oracle import reference_loader as rl

BF16 = torch.bfloat16


def test_h16_constants_and_round_trip_agree_between_product_and_reference_double():
    ops = sub("ops")
    assert ops.H16 == H16 == torch.float16 and ops.H16_SCALE == H16_SCALE == 2.0 ** -6
    assert (ops.STORE_BF16, ops.STORE_FP32, ops.STORE_H16) == (0, 1, 2)
    x = torch.randn(4096) * torch.logspace(-4, 5, 4096)
    h = _st(x, torch.empty(1, dtype=H16))
    back = _ld(h)
    assert h.dtype == H16 and torch.equal(back, ops.h16_to_float(h))
    big = x.abs() > 1e-2                                            # normal range of the shifted half: 11 significant bits
    assert float(((back - x).abs() / x.abs())[big].max()) <= 2.0 ** -11 * 1.0001
    assert float((back - x).abs()[~big].max()) <= 2.0 ** -24 / H16_SCALE * 0.5001 + 2.0 ** -11 * 1e-2   # absolute floor below it
    assert torch.isfinite(_ld(_st(torch.tensor([4.0e6, -4.0e6]), h))).all()                 # range +-4.19e6
    with pytest.raises(ValueError):
        ops.store_kind(torch.empty(1, dtype=torch.float64))


def test_vae_engine_storage_regimes_on_the_cpu_double():
    """trunk / branch kinds are an engine option; h16 tracks fp32 storage an order of magnitude closer than bf16 does, and the
    exact-arithmetic double (act_dtype fp32) ignores the option's default (host-logic tests stay exact)."""
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    cfg = config.VAE_TINY
    sd = weights.synth_vae_state_dict(cfg, seed=2)
    z = (torch.randn(2, 6, 8, 16, generator=torch.Generator().manual_seed(1))).to(BF16)
    run = lambda **kw: vae.VideoVAEEngine(cfg, sd, TorchOps("cpu", act_dtype=BF16), **kw).decode(z).float()
    exact_out = vae.VideoVAEEngine(cfg, sd, TorchOps("cpu", act_dtype=torch.float32)).decode(z.float()).float()   # no rounding anywhere
    e = {k: rel_err(run(trunk_store=k, branch_store=k), exact_out) for k in ("fp32", "h16", "bf16")}
    print("decode error against exact arithmetic, by storage of trunk and conv1 outputs:", {k: f"{v:.2e}" for k, v in e.items()})
    # (a random-weight VAE amplifies any perturbation to the operand-rounding floor, so the regimes are compared against the exact
    # result, not with each other: h16 sits with fp32 storage, bf16 storage is clearly worse)
    assert e["h16"] < 1.1 * e["fp32"] and e["bf16"] > 1.15 * e["h16"]
    eng = vae.VideoVAEEngine(cfg, sd, TorchOps("cpu", act_dtype=BF16))
    assert (eng.trunk_store, eng.branch_store) == ("h16", "h16") and eng.tile_streams == 1
    exact = vae.VideoVAEEngine(cfg, sd, TorchOps("cpu", act_dtype=torch.float32))
    assert (exact.trunk_store, exact.branch_store) == ("bf16", "bf16") and exact.trunk_dtype is None
    assert vae.VideoVAEEngine(cfg, sd, TorchOps("cpu", act_dtype=BF16), trunk_fp32=True).trunk_store == "fp32"       # rounds 2-3 spelling
    assert vae.VideoVAEEngine(cfg, sd, TorchOps("cpu", act_dtype=BF16), trunk_fp32=False).branch_store == "bf16"


def test_vae_engine_h16_overflow_guard_repeats_the_call_with_fp32_stores():
    """h16 ends at +-4.2e6.  A decoder whose conv_in is scaled so that the trunk exceeds that (bf16 / fp32, what the reference
    computes in, hold it easily) turns into inf -> NaN under h16 stores; the engine notices the non-finite result and runs the
    call again with fp32 stores: same answer as an engine built with fp32 stores, one warning, h16 restored afterwards."""
    config, weights, vae = sub("config"), sub("weights"), sub("vae")
    cfg = config.VAE_TINY
    sd = dict(weights.synth_vae_state_dict(cfg, seed=2))
    for k in ("decoder.conv_in.weight", "decoder.conv_in.bias", "encoder.conv_in.weight", "encoder.conv_in.bias"):
        sd[k] = sd[k] * 3.0e8
    z = (torch.randn(2, 6, 8, 16, generator=torch.Generator().manual_seed(1))).to(BF16)
    wide = vae.VideoVAEEngine(cfg, sd, TorchOps("cpu", act_dtype=BF16), trunk_store="fp32", branch_store="fp32").decode(z)
    assert torch.isfinite(wide.float()).all()
    unguarded = vae.VideoVAEEngine(cfg, sd, TorchOps("cpu", act_dtype=BF16), overflow_guard=False)
    assert not torch.isfinite(unguarded.decode(z).float()).all()             # the failure the guard exists for
    eng = vae.VideoVAEEngine(cfg, sd, TorchOps("cpu", act_dtype=BF16))
    with pytest.warns(RuntimeWarning, match="h16"):
        got = eng.decode(z)
    assert torch.equal(got, wide) and eng.overflow_reruns == 1
    assert (eng.trunk_store, eng.branch_store, eng.trunk_dtype, eng.branch_dtype) == ("h16", "h16", H16, H16)
    x = (torch.rand(3, 5, 16, 16, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(BF16)
    with pytest.warns(RuntimeWarning, match="h16"):
        lat = eng.encode(x)
    assert torch.isfinite(lat.float()).all() and eng.overflow_reruns == 2
    # ordinary weights: no rerun, no warning, no change of the result
    sd0 = weights.synth_vae_state_dict(cfg, seed=2)
    e0 = vae.VideoVAEEngine(cfg, sd0, TorchOps("cpu", act_dtype=BF16))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        a = e0.decode(z)
    assert e0.overflow_reruns == 0 and torch.equal(a, vae.VideoVAEEngine(cfg, sd0, TorchOps("cpu", act_dtype=BF16), overflow_guard=False).decode(z))
    # a non-finite INPUT (NaN latents from upstream) is not an h16 overflow: no fp32 re-run, the warning names the input
    zn = z.clone()
    zn[0, 0, 0, 0] = float("nan")
    with pytest.warns(RuntimeWarning, match="input of this call is not finite"):
        bad = e0.decode(zn)
    assert e0.overflow_reruns == 0 and not torch.isfinite(bad.float()).all()


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs the reference checkout (the recipe compiles it)")
def test_oracle_ref_recipe_compiles_the_reference_without_copying_source(tmp_path):
    """oracle/build_ref.py: a sourceless tree of byte-compiled modules + marshalled definitions; the loader imports the
    reference's NaDiT from it in a fresh interpreter that cannot see the checkout."""
    from oracle import build_ref
    out = build_ref.build(out_dir=str(tmp_path / "_ref"), verbose=False)
    files = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs]
    assert files and not [f for f in files if f.endswith(".py")]                  # no source text anywhere
    assert all(f.endswith((".pyc", ".marshal", "MANIFEST.json")) for f in files)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle import reference_loader as rl\n"
            "assert rl.available() and rl.kind() == 'compiled', (rl.REFERENCE_ROOT, rl.kind())\n"
            "cls = rl.reference_nadit_class(); ns = rl.reference_glue()\n"
            "print(cls.__name__, callable(ns['pad_video_temporal']))\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SEEDVR2_REFERENCE_ROOT=out), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and r.stdout.split() == ["NaDiT", "True"], r.stderr[-2000:]


def test_dit_engine_h16_stream_overflow_guard():
    """The NaDiT's residual stream in h16 (round 5) ends at +-4.2e6: an engine whose patch-in projection is scaled beyond that gets
    inf in the stream -> NaN out; forward() notices, repeats the call with an fp32 stream (same answer as an fp32-stream engine),
    warns once and keeps h16 for the next call.  A non-finite INPUT is not blamed on h16."""
    config, weights, dit = sub("config"), sub("weights"), sub("dit")
    cfg = config.DIT_TINY
    sd = dict(weights.synth_dit_state_dict(cfg, seed=3))
    sd["vid_in.proj.weight"] = sd["vid_in.proj.weight"] * 3.0e8
    g = torch.Generator().manual_seed(2)
    vid = torch.randn(1, 8, 12, 33, generator=g).to(BF16)
    txt = weights.synth_text_embedding()
    wide = dit.NaDiTEngine(cfg, sd, TorchOps("cpu", act_dtype=BF16), hid_store="fp32").forward(vid, txt, 1000.0)
    assert torch.isfinite(wide.float()).all()
    eng = dit.NaDiTEngine(cfg, sd, TorchOps("cpu", act_dtype=BF16))
    assert eng.hid_store == "h16"                                     # the default on a bf16 backend
    with pytest.warns(RuntimeWarning, match="h16 residual stream"):
        got = eng.forward(vid, txt, 1000.0)
    assert torch.equal(got, wide) and eng.overflow_reruns == 1 and eng.hid_store == "h16" and eng.hid_dtype == H16
    bad = vid.clone()
    bad[0, 0, 0, 0] = float("nan")
    with pytest.warns(RuntimeWarning, match="input of this call is not finite"):
        out = eng.forward(bad, txt, 1000.0)
    assert eng.overflow_reruns == 1 and not torch.isfinite(out.float()).all()
    # the exact-arithmetic CPU double keeps its one dtype whatever is asked for
    assert dit.NaDiTEngine(cfg, sd, TorchOps("cpu", act_dtype=torch.float32), hid_store="h16").hid_store == "fp32"
