"""Phase glue (transforms.py, colorfix.py, pipeline.py): host/device plumbing checked on CPU.

* against the reference's own function text where /root/reference is mounted (compiled unmodified by
  oracle/reference_loader.reference_glue()) -- bit-exact for index/padding logic, <= 1e-6 for fp32 colour maths;
* size-independent properties everywhere (4n+1 lengths, mirror structure, blend weights, batch plans cover
  every frame once, an identity runner reproduces the transformed input through all four phases).
"""
import math

import pytest
import torch

from conftest import sub
from oracle import reference_loader as rl

needs_ref = pytest.mark.skipif(not rl.available(), reason="/root/reference not mounted")


@pytest.fixture(scope="module")
def ref():
    return rl.reference_glue()


# ------------------------------------------------------------------ temporal padding / blending
@needs_ref
@pytest.mark.parametrize("t,count,dim,prepend", [(5, 0, 0, False), (6, 0, 1, False), (8, 0, 0, False), (2, 0, 1, False),
                                                 (1, 0, 0, False), (7, 3, 0, True), (3, 5, 0, True), (3, 7, 1, False),
                                                 (1, 4, 0, False), (9, 2, 1, True)])
def test_pad_video_temporal_equals_reference(ref, t, count, dim, prepend):
    tr = sub("transforms")
    shape = (t, 3, 4, 5) if dim == 0 else (3, t, 4, 5)
    x = torch.arange(math.prod(shape), dtype=torch.float32).reshape(shape)
    want = ref["pad_video_temporal"](x, count=count, temporal_dim=dim, prepend=prepend)
    assert torch.equal(tr.pad_video_temporal(x, count=count, temporal_dim=dim, prepend=prepend), want)


@pytest.mark.parametrize("t", range(1, 14))
def test_4n1_padding_properties(t):
    tr = sub("transforms")
    x = torch.arange(t, dtype=torch.float32).reshape(t, 1, 1, 1)
    y = tr.pad_video_temporal(x, temporal_dim=0)
    assert y.shape[0] % 4 == 1 and y.shape[0] - t < 4 and torch.equal(y[:t], x)
    pad = y[t:, 0, 0, 0].tolist()
    if t > len(pad):                                    # mirrored frames t-2, t-3, ...
        assert pad == [float(t - 2 - i) for i in range(len(pad))]


@needs_ref
@pytest.mark.parametrize("overlap", [1, 2, 3, 4, 7])
def test_blend_equals_reference(ref, overlap):
    tr = sub("transforms")
    g = torch.Generator().manual_seed(overlap)
    a, b = torch.rand(overlap, 6, 5, 3, generator=g), torch.rand(overlap, 6, 5, 3, generator=g)
    assert torch.equal(tr.blend_overlapping_frames(a, b, overlap), ref["blend_overlapping_frames"](a, b, overlap))
    assert torch.allclose(tr.blend_overlapping_frames(a, a, overlap), a, atol=1e-6)      # weights sum to one


# ------------------------------------------------------------------ input transform
@needs_ref
@pytest.mark.parametrize("h,w,res,mx", [(120, 212, 256, 0), (478, 320, 400, 0), (90, 160, 720, 1000), (64, 64, 100, 0),
                                         (100, 37, 64, 150)])
def test_video_transform_equals_reference(ref, h, w, res, mx):
    tr = sub("transforms")
    g = torch.Generator().manual_seed(h + w)
    x = torch.rand(3, 3, h, w, generator=g)
    want = ref["SideResize"](size=res, max_size=mx)(x)
    got = tr.side_resize(x, res, mx)
    assert got.shape == want.shape and torch.equal(got, want)
    th, tw = tr.true_target_dims(h, w, res, mx)
    assert (th, tw) == ((want.shape[-2] // 2) * 2, (want.shape[-1] // 2) * 2)
    padded = ref["DivisiblePad"]((16, 16))(torch.clamp(want, 0.0, 1.0))
    full = tr.video_transform(x, res, mx)
    assert torch.equal(full, ((padded - 0.5) / 0.5).permute(1, 0, 2, 3))
    assert full.shape[-1] % 16 == 0 and full.shape[-2] % 16 == 0 and float(full.min()) >= -1 and float(full.max()) <= 1


def test_resized_output_size_matches_interpolate_shapes():
    tr = sub("transforms")
    assert tr.resized_output_size(720, 1280, 2160) == (2160, 3840)
    assert tr.resized_output_size(1280, 720, 2160) == (3840, 2160)
    assert tr.resized_output_size(478, 320, 400) == (597, 400)          # int(400 * 478 / 320) = 597
    assert tr.true_target_dims(478, 320, 400) == (596, 400)


# ------------------------------------------------------------------ colour correction
@needs_ref
@pytest.mark.parametrize("method", ["adain", "wavelet", "lab", "hsv", "wavelet_adaptive"])
def test_colorfix_equals_reference(ref, method):
    cf = sub("colorfix")
    g = torch.Generator().manual_seed(7)
    content = (torch.rand(2, 3, 48, 72, generator=g) * 2 - 1) * 0.9
    style = (torch.rand(2, 3, 48, 72, generator=g) * 2 - 1) * 0.6 + 0.1
    if method in ("hsv", "wavelet_adaptive"):
        class _Dbg:
            def log(self, *a, **k):
                pass
        content[:, :, :8, :8] = 0.3                     # grey block: the zero-range (hue 0, saturation 0) branch
        content[:, 0, 8:16] = content[:, 1, 8:16]       # r == g ties
        fn = ref["hsv_saturation_histogram_match" if method == "hsv" else "wavelet_adaptive_color_correction"]
        want = fn(content.clone(), style.clone(), _Dbg())
    elif method == "adain":
        want = ref["adaptive_instance_normalization"](content.clone(), style.clone())
    elif method == "wavelet":
        want = ref["wavelet_reconstruction"](content.clone(), style.clone())
    else:
        class _Dbg:
            def log(self, *a, **k):
                pass
        want = ref["lab_color_transfer"](content.clone(), style.clone(), _Dbg(), luminance_weight=0.8)
    got = cf.METHODS[method](content, style)
    assert got.shape == want.shape
    if method != "lab":
        assert float((got - want).abs().max()) < 2e-6
        return
    # LAB: the histogram matching is RANK based, and its input is the wavelet base.  The blur here is the reference's dilated
    # 3x3 conv evaluated as two three-tap passes (same weights, another order of fp32 additions: <= 2e-6, the "wavelet"
    # case above), so a few near-tied pixels swap ranks and exchange their matched values.  Hence: (a) everything after
    # the blur is exact when it is given the reference's wavelet base, (b) composed, all but a handful of values agree.
    monkey = pytest.MonkeyPatch()
    try:
        monkey.setattr(cf, "wavelet_reconstruction", lambda c, s: ref["wavelet_reconstruction"](c.clone(), s.clone(), _Dbg()))
        staged = cf.lab_color_transfer(content, style, luminance_weight=0.8)
    finally:
        monkey.undo()
    assert float((staged - want).abs().max()) < 2e-6
    d = (got - want).abs()
    assert float((d > 2e-6).float().mean()) < 5e-3 and float(d.max()) < 1e-3


def test_colorfix_properties():
    cf = sub("colorfix")
    g = torch.Generator().manual_seed(3)
    x = torch.rand(1, 3, 40, 40, generator=g) * 2 - 1
    hi, lo = cf.wavelet_decomposition(x)
    assert torch.allclose(hi + lo, x, atol=1e-5)                       # the pyramid is a partition of the signal
    assert torch.allclose(cf.wavelet_reconstruction(x, x), x.clamp(-1, 1), atol=1e-5)
    # the blur = the reference's conv: depthwise 3x3 binomial kernel, dilation r, replicate padding (color_fix.py:122-157),
    # for every dilation of the pyramid, odd sizes, images smaller than the dilation cap, and bf16 storage (one rounding of
    # the fp32 sum, as a conv with fp32 accumulation gives)
    F = torch.nn.functional
    k = torch.tensor([[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]])[None, None].repeat(3, 1, 1, 1)
    for hh, ww in ((40, 40), (37, 131), (5, 9), (136, 129)):
        img = torch.rand(2, 3, hh, ww, generator=g) * 2 - 1
        for r in (1, 2, 4, 8, 16):
            rr = min(r, max(1, min(hh, ww) // 8))
            want = F.conv2d(F.pad(img, (rr, rr, rr, rr), mode="replicate"), k, groups=3, dilation=rr)
            got = cf.wavelet_blur(img, r)
            assert got.shape == img.shape and got.dtype == img.dtype and float((got - want).abs().max()) < 5e-7
            got16 = cf.wavelet_blur(img.bfloat16(), r)
            want16 = F.conv2d(F.pad(img.bfloat16().float(), (rr, rr, rr, rr), mode="replicate"), k, groups=3, dilation=rr).bfloat16()
            assert got16.dtype == torch.bfloat16 and float((got16 != want16).float().mean()) < 2e-3
            assert float((got16.float() - want16.float()).abs().max()) <= 2.0 ** -8
    rgb = torch.rand(2, 3, 16, 16, generator=g)
    assert torch.allclose(cf.lab_to_rgb(cf.rgb_to_lab(rgb)), rgb, atol=2e-4)             # LAB round trip
    a, b = torch.randn(1000, generator=g), torch.randn(1000, generator=g) * 3 + 1
    m = cf.histogram_match(a, b)
    assert torch.equal(torch.sort(m).values, torch.sort(b).values)     # exact quantile mapping, order preserved
    assert torch.equal(torch.argsort(m), torch.argsort(a))
    same = cf.lab_color_transfer(x, x, luminance_weight=0.8)
    assert float((same - cf.wavelet_reconstruction(x, x)).abs().max()) < 5e-3           # matching to itself is ~identity


# ------------------------------------------------------------------ batch planning + the four phases
@pytest.mark.parametrize("total,bs,ov,uniform", [(33, 33, 0, False), (128, 17, 0, True), (20, 9, 3, True), (7, 5, 2, False),
                                                  (5, 5, 4, False), (11, 4, 0, False), (6, 3, 5, False)])
def test_plan_batches_covers_every_frame_once(total, bs, ov, uniform):
    pl = sub("pipeline")
    plans, ov_used = pl.plan_batches(total, bs, ov, uniform)
    written = 0
    for i, p in enumerate(plans):
        ori = p.end - p.start
        assert 0 < ori <= bs and (p.uniform_pad == (bs - ori if uniform else 0))
        assert p.start == (0 if i == 0 else written - ov_used)
        written += ori if i == 0 or ov_used == 0 else ori - ov_used
    assert written == total


class _IdentityRunner:
    """vae_decode(vae_encode(x)) == x, DiT = identity on the latent carried in the condition: lets the bookkeeping
    of all four phases (batching, padding, trims, overlap blend, colour-fix windows) be checked exactly on CPU."""

    class _Dev:
        device = torch.device("cpu")

    def __init__(self):
        self.dit, self.store = self._Dev(), {}
        self.schedule = sub("runner").LinearInterpolationSchedule(1000.0)

    def timestep_transform(self, t, shape):
        return t

    def vae_encode(self, xs):
        out = []
        for x in xs:
            c, t, h, w = x.shape
            lat = torch.zeros((t - 1) // 4 + 1, h // 8, w // 8, 16, dtype=x.dtype)
            lat[0, 0, 0, 0] = float(len(self.store) + 1)
            self.store[len(self.store) + 1] = x
            out.append(lat)
        return out

    def get_condition(self, noise, task, latent_blur):
        return torch.cat([latent_blur, torch.ones_like(latent_blur[..., :1])], dim=-1)

    def inference(self, noises, conditions, texts_pos, texts_neg):
        return [c[..., :16] for c in conditions]

    def vae_decode(self, lats):
        return [self.store[int(round(float(l[0, 0, 0, 0])))] for l in lats]


@pytest.mark.parametrize("total,bs,ov,uniform,prepend", [(9, 9, 0, False, 0), (13, 5, 0, True, 0), (14, 6, 2, False, 0),
                                                          (10, 5, 3, True, 2), (3, 8, 0, False, 0)])
def test_pipeline_identity_runner_reproduces_transformed_input(total, bs, ov, uniform, prepend):
    pl, tr = sub("pipeline"), sub("transforms")
    g = torch.Generator().manual_seed(total)
    images = torch.rand(total, 20, 28, 3, generator=g)
    out = pl.upscale(images, _IdentityRunner(), torch.zeros(58, 8), resolution=40, batch_size=bs, uniform_batch_size=uniform,
                     temporal_overlap=ov, prepend_frames=prepend, color_correction="none")
    th, tw = tr.true_target_dims(20, 28, 40)
    want = tr.side_resize(images.permute(0, 3, 1, 2), 40).clamp(0, 1)[:, :, :th, :tw].permute(0, 2, 3, 1)
    assert out.shape == (total, th, tw, 3) and out.dtype == torch.float32   # (the decoded frames are held in fp32: ComfyUI's IMAGE dtype)
    assert float((out.float() - want).abs().max()) < 1.2e-2              # bf16 storage of the [-1, 1] input / decoder output
    out_bf = pl.upscale(images, _IdentityRunner(), torch.zeros(58, 8), resolution=40, batch_size=bs, uniform_batch_size=uniform,
                        temporal_overlap=ov, prepend_frames=prepend, color_correction="none", output_dtype=None)
    assert out_bf.dtype == torch.bfloat16 and float((out_bf.float() - want).abs().max()) < 1.2e-2      # rounds 2-3: storage dtype throughout
    # colour correction against the input itself must keep the identity result (within bf16)
    out2 = pl.upscale(images, _IdentityRunner(), torch.zeros(58, 8), resolution=40, batch_size=bs, uniform_batch_size=uniform,
                      temporal_overlap=ov, prepend_frames=prepend, color_correction="wavelet")
    assert float((out2.float() - want).abs().max()) < 3e-2


def test_pipeline_batch_filter_partitions_the_output():
    """Data parallelism over temporal batches: the per-rank outputs are disjoint and sum to the single-rank result."""
    pl = sub("pipeline")
    images = torch.rand(19, 16, 16, 3, generator=torch.Generator().manual_seed(1))
    kw = dict(resolution=32, batch_size=5, uniform_batch_size=True, color_correction="adain")
    full = pl.upscale(images, _IdentityRunner(), torch.zeros(58, 8), **kw)
    parts = [pl.upscale(images, _IdentityRunner(), torch.zeros(58, 8), batch_filter=lambda i, r=r: i % 2 == r, **kw) for r in (0, 1)]
    assert torch.equal(parts[0] + parts[1], full)
    assert float((parts[0] * parts[1]).abs().max()) == 0.0


def test_transformed_shape_and_phase1_noise_stream():
    """pipeline.transformed_shape predicts prepare_batch's result shape (what a rank draws-and-discards for the batches
    it skips), and phase 1's input noise is a function of ``seed`` alone (seed + 1e6 before the first batch,
    generation_phases.py:327-330), not of whatever the global generator held."""
    pipeline = sub("pipeline")
    g = torch.Generator().manual_seed(2)
    images = torch.rand(13, 20, 36, 3, generator=g)
    plans, _ = pipeline.plan_batches(13, 5, 1, True)
    for res, mx in ((40, 0), (64, 100), (30, 0)):
        for plan in plans:
            assert tuple(pipeline.prepare_batch(images, plan, res, mx).shape) == pipeline.transformed_shape(images, plan, res, mx)
    kw = dict(resolution=32, batch_size=5, temporal_overlap=1, color_correction="none", input_noise_scale=0.8, seed=7)
    torch.manual_seed(123)
    a = pipeline.upscale(images, _IdentityRunner(), torch.zeros(58, 8), **kw)
    torch.manual_seed(456)
    b = pipeline.upscale(images, _IdentityRunner(), torch.zeros(58, 8), **kw)
    c = pipeline.upscale(images, _IdentityRunner(), torch.zeros(58, 8), **{**kw, "seed": 8})
    assert torch.equal(a, b) and not torch.equal(a, c)
