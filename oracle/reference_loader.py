"""Import the reference's own model code, UNMODIFIED, from /root/reference
(test infrastructure only; /root/reference exists in the build container, not
on the GPU box -- callers must check ``available()`` first).

Reference entry points wrapped here:
  NaDiT                         src/models/dit_3b/nadit.py:39
  VideoAutoencoderKLWrapper     src/models/video_vae_v3/modules/attn_video_vae.py:1660
"""
import contextlib
import io
import os
import sys

REFERENCE_ROOT = os.environ.get("SEEDVR2_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "models", "dit_3b"))


def _prepare():
    from . import third_party_shims
    third_party_shims.install()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def reference_nadit_class():
    _prepare()
    with contextlib.redirect_stdout(io.StringIO()):
        from src.models.dit_3b.nadit import NaDiT
    return NaDiT


def reference_window_module():
    _prepare()
    with contextlib.redirect_stdout(io.StringIO()):
        from src.models.dit_3b import window
    return window


def reference_vae_class():
    _prepare()
    with contextlib.redirect_stdout(io.StringIO()):
        from src.models.video_vae_v3.modules.attn_video_vae import VideoAutoencoderKLWrapper
    return VideoAutoencoderKLWrapper


def build_reference_dit(cfg: dict, state_dict=None, dtype=None):
    """cfg uses the keys of configs_3b/main.yaml:11-36 (see dit_config() in the package)."""
    import torch
    NaDiT = reference_nadit_class()
    L = cfg["num_layers"]
    with torch.device("meta") if state_dict is not None else contextlib.nullcontext():
        m = NaDiT(
            vid_in_channels=cfg["vid_in_channels"], vid_out_channels=cfg["vid_out_channels"],
            vid_dim=cfg["vid_dim"], vid_out_norm="fusedrms", txt_in_dim=cfg["txt_in_dim"],
            txt_in_norm="fusedln", txt_dim=cfg["vid_dim"], emb_dim=6 * cfg["vid_dim"],
            heads=cfg["heads"], head_dim=cfg["head_dim"], expand_ratio=4, norm="fusedrms",
            norm_eps=cfg["norm_eps"], ada="single", qk_bias=False, qk_norm="fusedrms",
            patch_size=[1, 2, 2], num_layers=L, mm_layers=cfg["mm_layers"], mlp_type="swiglu",
            msa_type=None, block_type=["mmdit_sr"] * L, window=[(4, 3, 3)] * L,
            window_method=(["720pwin_by_size_bysize", "720pswin_by_size_bysize"] * L)[:L],
            rope_type="mmrope3d", rope_dim=128,
        )
    if state_dict is not None:
        m.load_state_dict(state_dict, strict=True, assign=True)
    m = m.eval().requires_grad_(False)
    if dtype is not None:
        m = m.to(dtype)
    return m


VAE_CFG = dict(  # src/models/video_vae_v3/s8_c16_t4_inflation_sd3.yaml
    act_fn="silu", block_out_channels=[128, 256, 512, 512],
    down_block_types=["DownEncoderBlock3D"] * 4, in_channels=3, latent_channels=16,
    layers_per_block=2, norm_num_groups=32, out_channels=3, slicing_sample_min_size=4,
    temporal_scale_num=2, inflation_mode="pad", up_block_types=["UpDecoderBlock3D"] * 4,
    spatial_downsample_factor=8, temporal_downsample_factor=4, use_quant_conv=False,
    use_post_quant_conv=False, freeze_encoder=False, gradient_checkpoint=True,
)


def build_reference_vae(state_dict=None, dtype=None, block_out_channels=None, slicing=True):
    import torch
    cls = reference_vae_class()
    cfg = dict(VAE_CFG)
    if block_out_channels is not None:
        cfg["block_out_channels"] = list(block_out_channels)
    with torch.device("meta") if state_dict is not None else contextlib.nullcontext():
        vae = cls(**cfg)
    if state_dict is not None:
        vae.load_state_dict(state_dict, strict=True, assign=True)
    vae = vae.eval().requires_grad_(False)
    if dtype is not None:
        vae = vae.to(dtype)
    vae.debug = None                      # normally set at model_configuration.py:1273-1274
    vae.tensor_offload_device = None
    if slicing:                            # configs_3b/main.yaml:53-58
        vae.set_causal_slicing(split_size=4, memory_device="same")
        vae.set_memory_limit(conv_max_mem=0.5, norm_max_mem=0.5)
    return vae
