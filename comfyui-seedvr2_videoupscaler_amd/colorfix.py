"""Colour correction after VAE decode: transfers the input clip's colours onto the
upscaled frames.  Restated from src/utils/color_fix.py (results identical on the same inputs):

  adaptive_instance_normalization   :72-120   per-channel mean / std transfer
  wavelet_blur / _decomposition     :122-185  5-level a-trous pyramid, 3x3 binomial kernel, dilation 2^i capped at
                                               min(H, W) // 8, replicate padding
  wavelet_reconstruction            :187-247  content high frequencies + style low frequencies, clamp [-1, 1]
  lab_color_transfer                :249-366  wavelet base -> CIELAB (D65) -> per-channel histogram matching
                                               (a*, b* fully, L* blended with luminance_weight) -> RGB
  _rgb_to_lab_batch / _lab_to_rgb_batch / _histogram_matching_channel   :368-522
  hsv_saturation_histogram_match    :524-612  HSV; saturation histogram-matched per 30-degree hue bin (bins with
                                               <= 100 pixels on either side are left alone), H and V kept
  _rgb_to_hsv_batch / _hsv_to_rgb_batch / _hue_conditional_saturation_match / _histogram_match_1d   :614-770
  wavelet_adaptive_color_correction :772-856  wavelet base, HSV result blended in where content is oversaturated
                                               (sigmoid(5 (dS - 0.15)), gated by the wavelet result's own excess)

All tensors are [B, C, H, W] in [-1, 1]; the work is HBM / sort bound (no MFMA), so it stays torch glue on
the device.
"""
import torch
import torch.nn.functional as F


def _resize_like(style: torch.Tensor, content: torch.Tensor) -> torch.Tensor:
    if style.shape[-2:] == content.shape[-2:]:
        return style
    return F.interpolate(style.float(), size=content.shape[-2:], mode="bilinear", align_corners=False).to(style.dtype)


def adaptive_instance_normalization(content: torch.Tensor, style: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    def mean_std(x):
        b, c = x.shape[:2]
        flat = x.reshape(b, c, -1)
        return flat.mean(dim=2).reshape(b, c, 1, 1), (flat.var(dim=2) + eps).sqrt().reshape(b, c, 1, 1)

    s_mean, s_std = mean_std(style)
    c_mean, c_std = mean_std(content)
    return (content - c_mean) / c_std * s_std + s_mean


def wavelet_blur(image: torch.Tensor, radius: int) -> torch.Tensor:
    """The reference's dilated depthwise 3x3 conv over the replicate-padded image (color_fix.py:122-157).  The binomial
    kernel is separable ([1, 2, 1] / 4 down the rows, then along them) and its weights are powers of two, so it is evaluated
    as two three-tap passes of shifted adds in fp32 with ONE rounding to the image dtype at the end -- the arithmetic of a
    conv with fp32 accumulation, without handing a [B, 3, H, W] depthwise conv with dilation up to 16 to a convolution
    library (on MI355X MIOpen serves it with its naive fallback kernel: 75 ms per 17-frame 4K batch and level, against a
    few HBM passes here)."""
    radius = min(radius, max(1, min(image.shape[-2:]) // 8))
    h, w = image.shape[-2:]
    x = image if image.dtype in (torch.float32, torch.float64) else image.float()
    rows = torch.arange(h, device=image.device)
    v = x.index_select(-2, (rows - radius).clamp_(min=0)) + x.index_select(-2, (rows + radius).clamp_(max=h - 1))
    v = v.mul_(0.25).add_(x, alpha=0.5)                                   # rows y - r, y, y + r (clamped = replicate padding)
    p = F.pad(v, (radius, radius, 0, 0), mode="replicate")
    out = (p[..., :w] + p[..., 2 * radius:2 * radius + w]).mul_(0.25).add_(v, alpha=0.5)
    return out.to(image.dtype)


def wavelet_decomposition(image: torch.Tensor, levels: int = 5):
    high = torch.zeros_like(image)
    low = image
    for i in range(levels):
        low = wavelet_blur(image, 2 ** i)
        high = high + image - low
        image = low
    return high, low


def wavelet_reconstruction(content: torch.Tensor, style: torch.Tensor) -> torch.Tensor:
    style = _resize_like(style, content)
    high, _ = wavelet_decomposition(content)
    _, low = wavelet_decomposition(style)
    return (high + low).clamp(-1.0, 1.0)


_RGB2XYZ = [[0.4124564, 0.3575761, 0.1804375], [0.2126729, 0.7151522, 0.0721750], [0.0193339, 0.1191920, 0.9503041]]
_XYZ2RGB = [[3.2404542, -1.5371385, -0.4985314], [-0.9692660, 1.8760108, 0.0415560], [0.0556434, -0.2040259, 1.0572252]]
_EPS, _KAPPA = 6.0 / 29.0, (29.0 / 3.0) ** 3


def _mix3(x: torch.Tensor, m) -> torch.Tensor:
    """[B, 3, H, W] -> per-pixel 3x3 colour matrix as the reference applies it: ``x.permute(0, 2, 3, 1) @ m.T`` on [N, 3] rows
    (color_fix.py:368-431).  Kept as the same fp32 matmul on purpose: the histogram matching that follows is rank based, so a
    1-ulp change here would swap the matched values of near-tied pixels."""
    b, _, h, w = x.shape
    mt = torch.tensor(m, dtype=torch.float32, device=x.device)
    return torch.matmul(x.permute(0, 2, 3, 1).reshape(-1, 3), mt.T).reshape(b, h, w, 3).permute(0, 3, 1, 2).clone()


def rgb_to_lab(rgb: torch.Tensor) -> torch.Tensor:
    """[B, 3, H, W] sRGB in [0, 1] (fp32) -> CIELAB, D65."""
    lin = torch.where(rgb > 0.04045, torch.pow((rgb + 0.055) / 1.055, 2.4), rgb / 12.92)
    xyz = _mix3(lin, _RGB2XYZ)
    xyz[:, 0] = xyz[:, 0] / 0.95047
    xyz[:, 2] = xyz[:, 2] / 1.08883
    f = torch.where(xyz > _EPS ** 3, torch.pow(xyz, 1.0 / 3.0), (xyz * _KAPPA + 16.0) / 116.0)
    return torch.stack([f[:, 1] * 116.0 - 16.0, (f[:, 0] - f[:, 1]) * 500.0, (f[:, 1] - f[:, 2]) * 200.0], dim=1)


def lab_to_rgb(lab: torch.Tensor) -> torch.Tensor:
    fy = (lab[:, 0] + 16.0) / 116.0
    fx = lab[:, 1] / 500.0 + fy
    fz = fy - lab[:, 2] / 200.0

    def inv(f):
        return torch.where(f > _EPS, torch.pow(f, 3.0), (f * 116.0 - 16.0) / _KAPPA)

    xyz = torch.stack([inv(fx) * 0.95047, inv(fy), inv(fz) * 1.08883], dim=1)
    lin = _mix3(xyz, _XYZ2RGB)
    rgb = torch.where(lin > 0.0031308, torch.pow(torch.clamp(lin, min=0.0), 1.0 / 2.4) * 1.055 - 0.055, lin * 12.92)
    return torch.clamp(rgb, 0.0, 1.0)


def histogram_match(source: torch.Tensor, reference: torch.Tensor) -> torch.Tensor:
    """Quantile mapping of ``source`` onto ``reference`` (any shapes): the k-th smallest source value becomes the
    k-th smallest reference value (nearest-rank when the sizes differ)."""
    shape = source.shape
    src_sorted_idx = torch.sort(source.flatten()).indices
    ref_sorted = torch.sort(reference.flatten()).values
    n_s, n_r = src_sorted_idx.numel(), ref_sorted.numel()
    if n_s != n_r:
        q = torch.linspace(0, 1, n_s, device=source.device)
        ref_sorted = ref_sorted[(q * (n_r - 1)).long().clamp_(0, n_r - 1)]
    out = torch.empty_like(ref_sorted)
    out[src_sorted_idx] = ref_sorted          # == matched_sorted[argsort(source_indices)]
    return out.reshape(shape)


def lab_color_transfer(content: torch.Tensor, style: torch.Tensor, luminance_weight: float = 0.8) -> torch.Tensor:
    content = wavelet_reconstruction(content, style)
    style = _resize_like(style, content)
    dt = content.dtype
    c = ((content.float() + 1.0) * 0.5).clamp(0.0, 1.0)
    s = ((style.float() + 1.0) * 0.5).clamp(0.0, 1.0)
    c_lab, s_lab = rgb_to_lab(c), rgb_to_lab(s)
    a = histogram_match(c_lab[:, 1], s_lab[:, 1])
    b = histogram_match(c_lab[:, 2], s_lab[:, 2])
    if luminance_weight < 1.0:
        L = c_lab[:, 0] * luminance_weight + histogram_match(c_lab[:, 0], s_lab[:, 0]) * (1.0 - luminance_weight)
    else:
        L = c_lab[:, 0]
    rgb = lab_to_rgb(torch.stack([L, a, b], dim=1))
    return (rgb * 2.0 - 1.0).to(dt)


def rgb_to_hsv(rgb: torch.Tensor) -> torch.Tensor:
    """[B, 3, H, W] in [0, 1] -> (h, s, v) in [0, 1]; ties between channels resolve blue > green > red."""
    r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    maxc, minc = rgb.max(dim=1).values, rgb.min(dim=1).values
    rng = maxc - minc
    colour = rng > 1e-10
    d = torch.where(colour, rng, torch.ones_like(rng))
    h = torch.where((maxc == r) & colour, torch.remainder((g - b) / d, 6.0), torch.zeros_like(maxc))
    h = torch.where((maxc == g) & colour, (b - r) / d + 2.0, h)
    h = torch.where((maxc == b) & colour, (r - g) / d + 4.0, h)
    s = torch.where(maxc > 1e-10, rng / maxc.clamp(min=1e-10), torch.zeros_like(maxc))
    return torch.stack([h / 6.0, s, maxc], dim=1)


def hsv_to_rgb(hsv: torch.Tensor) -> torch.Tensor:
    h6, s, v = hsv[:, 0] * 6.0, hsv[:, 1], hsv[:, 2]
    sector = torch.floor(h6).long() % 6
    f = h6 - torch.floor(h6)
    p, q, t = v * (1.0 - s), v * (1.0 - s * f), v * (1.0 - s * (1.0 - f))
    table = ((v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q))     # (r, g, b) of sectors 0..5
    out = []
    for ch in range(3):
        x = torch.zeros_like(v)
        for k in range(6):
            x = torch.where(sector == k, table[k][ch], x)
        out.append(x)
    return torch.stack(out, dim=1)


def hue_conditional_saturation_match(c_h, c_s, s_h, s_s, bins: int = 12, min_pixels: int = 100) -> torch.Tensor:
    """Saturation of the content pixels of each hue bin histogram-matched to the style pixels of the same bin (bin 0
    also collects h >= 1 - 1/bins: the red wrap-around, so those pixels may be matched twice, last bin winning)."""
    width = 1.0 / bins
    out = c_s.clone()
    for k in range(bins):
        lo, hi = k * width, (k + 1) * width
        if k == 0:
            cm = ((c_h >= 0) & (c_h < hi)) | (c_h >= 1.0 - width)
            sm = ((s_h >= 0) & (s_h < hi)) | (s_h >= 1.0 - width)
        else:
            cm, sm = (c_h >= lo) & (c_h < hi), (s_h >= lo) & (s_h < hi)
        cs, ss = c_s[cm], s_s[sm]
        if cs.numel() > min_pixels and ss.numel() > min_pixels:
            out[cm] = histogram_match(cs, ss)
    return out


def hsv_saturation_histogram_match(content: torch.Tensor, style: torch.Tensor) -> torch.Tensor:
    style = _resize_like(style, content)
    dt = content.dtype
    c = rgb_to_hsv(((content.float() + 1.0) * 0.5).clamp(0.0, 1.0))
    s = rgb_to_hsv(((style.float() + 1.0) * 0.5).clamp(0.0, 1.0))
    sat = hue_conditional_saturation_match(c[:, 0], c[:, 1], s[:, 0], s[:, 1])
    rgb = hsv_to_rgb(torch.stack([c[:, 0], sat, c[:, 2]], dim=1)).clamp(0.0, 1.0)
    return (rgb * 2.0 - 1.0).to(dt)


def saturation_map(x: torch.Tensor) -> torch.Tensor:
    rgb = ((x + 1.0) * 0.5).clamp(0.0, 1.0)
    maxc, minc = rgb.max(dim=1, keepdim=True).values, rgb.min(dim=1, keepdim=True).values
    return torch.where(maxc > 1e-10, (maxc - minc) / maxc.clamp(min=1e-10), torch.zeros_like(maxc))


def wavelet_adaptive_color_correction(content: torch.Tensor, style: torch.Tensor, threshold: float = 0.15,
                                      sharpness: float = 5.0) -> torch.Tensor:
    style = _resize_like(style, content)
    dt = content.dtype
    content, style = content.float(), style.float()
    base = wavelet_reconstruction(content, style)
    hsv = hsv_saturation_histogram_match(content, style)
    s_sat = saturation_map(style)
    w = torch.sigmoid(sharpness * (saturation_map(content) - s_sat - threshold))
    w = (w * ((saturation_map(base) - s_sat) > threshold * 0.5).float()).clamp(0.0, 1.0)
    return (base * (1.0 - w) + hsv * w).to(dt)


METHODS = {
    "lab": lambda c, s: lab_color_transfer(c, s, luminance_weight=0.8),
    "wavelet_adaptive": wavelet_adaptive_color_correction,
    "hsv": hsv_saturation_histogram_match,
    "wavelet": wavelet_reconstruction,
    "adain": adaptive_instance_normalization,
}
