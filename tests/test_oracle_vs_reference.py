"""Live pin of the oracle + the package's window planner against the reference implementation.
Runs only where /root/reference is mounted (the build container)."""
import pytest
import torch

from conftest import sub, rel_err
from oracle import reference_loader as rl
from oracle import dit_oracle, vae_oracle

pytestmark = pytest.mark.skipif(not rl.available(), reason="/root/reference not mounted")


@pytest.mark.parametrize("size", [(1, 16, 16), (3, 128, 128), (9, 135, 240), (5, 18, 30), (17, 135, 240),
                                  (2, 7, 11), (31, 40, 23), (4, 45, 80)])
def test_window_boxes_equal_reference(size):
    windows = sub("windows")
    ref = rl.reference_window_module()
    for method, fn in ((windows.REGULAR, ref.make_720Pwindows_bysize),
                       (windows.SHIFTED, ref.make_shifted_720Pwindows_bysize)):
        want = [(a.start, a.stop, b.start, b.stop, c.start, c.stop) for a, b, c in fn(size, (4, 3, 3))]
        assert windows.window_boxes(size, (4, 3, 3), method) == want


def test_dit_tiny_oracle_equals_reference_ragged():
    config, weights, windows = sub("config"), sub("weights"), sub("windows")
    cfg = config.DIT_TINY
    sd = weights.synth_dit_state_dict(cfg, seed=5)
    ref = rl.build_reference_dit(cfg.as_dict(), {k: v.float() for k, v in sd.items()})
    torch.manual_seed(1)
    T, H, W = 5, 36, 60
    vid = torch.randn(T, H, W, 33)
    txt = torch.randn(58, 5120)
    with torch.no_grad():
        want = ref(vid=vid.reshape(-1, 33), txt=txt, vid_shape=torch.tensor([[T, H, W]]),
                   txt_shape=torch.tensor([[58]]), timestep=torch.tensor([1000.0])).vid_sample.reshape(T, H, W, 16)
    got = dit_oracle.dit_forward(sd, cfg, vid, txt, 1000.0, windows_mod=windows)
    assert rel_err(got, want) < 2e-5


def test_vae_oracle_equals_reference_sliced():
    config, weights = sub("config"), sub("weights")
    cfg = config.VAE_V3
    sd = weights.synth_vae_state_dict(cfg, seed=9)
    ref = rl.build_reference_vae({k: v.float() for k, v in sd.items()})   # slicing + 0.5 GiB limits on
    torch.manual_seed(2)
    x = torch.rand(1, 3, 9, 32, 48) * 2 - 1
    z = torch.randn(1, 16, 3, 4, 6)
    with torch.no_grad():
        assert rel_err(vae_oracle.encode(x, sd, cfg), ref.encode(x).latent) < 2e-5
        assert rel_err(vae_oracle.decode(z, sd, cfg), ref.decode(z).sample) < 2e-5


def test_window_boxes_equal_reference_fuzz():
    """Randomised sweep of the window planner against the reference's window.py (both families, several window counts)."""
    import random
    windows = sub("windows")
    ref = rl.reference_window_module()
    rnd = random.Random(1234)
    for _ in range(150):
        size = (rnd.randint(1, 40), rnd.randint(1, 140), rnd.randint(1, 250))
        num = rnd.choice([(4, 3, 3), (1, 3, 3), (2, 2, 2), (4, 4, 4)])
        for method, fn in ((windows.REGULAR, ref.make_720Pwindows_bysize),
                           (windows.SHIFTED, ref.make_shifted_720Pwindows_bysize)):
            want = [(a.start, a.stop, b.start, b.stop, c.start, c.stop) for a, b, c in fn(size, num)]
            assert windows.window_boxes(size, num, method) == want, (size, num, method)


def test_temporal_padding_and_blend_equal_reference_fuzz():
    """pad_video_temporal / blend_overlapping_frames against the reference's function text over random lengths."""
    import random
    tr = sub("transforms")
    ns = rl.reference_glue()
    rnd = random.Random(7)
    for _ in range(120):
        t = rnd.randint(1, 23)
        x = torch.randn(t, 2, 3, 3)
        count = rnd.choice([0, 0, rnd.randint(1, 30)])
        prepend = count > 0 and rnd.random() < 0.4
        got = tr.pad_video_temporal(x, count=count, temporal_dim=0, prepend=prepend)
        want = ns["pad_video_temporal"](x.clone(), count=count, temporal_dim=0, prepend=prepend)
        assert got.shape == want.shape and torch.equal(got, want), (t, count, prepend)
    for ov in range(1, 9):
        a, b = torch.randn(ov, 4, 5, 3), torch.randn(ov, 4, 5, 3)
        assert torch.equal(tr.blend_overlapping_frames(a, b, ov), ns["blend_overlapping_frames"](a.clone(), b.clone(), ov))
