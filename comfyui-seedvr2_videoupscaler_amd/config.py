"""Model hyper-parameters of the SeedVR2 hot path, transcribed from the reference YAMLs
(omegaconf is not available here; values cited per key).

  configs_3b/main.yaml:11-36            -> DIT_3B
  configs_7b/main.yaml:11-33            -> DIT_7B (shape constants only; 7B path is a later row)
  video_vae_v3/s8_c16_t4_inflation_sd3.yaml, configs_3b/main.yaml:45-63 -> VAE_V3
"""
from dataclasses import dataclass, field, asdict
from typing import Tuple


@dataclass(frozen=True)
class DiTConfig:
    vid_in_channels: int = 33        # main.yaml:11  (16 noise + 16 cond + 1 mask)
    vid_out_channels: int = 16
    vid_dim: int = 2560
    txt_in_dim: int = 5120
    heads: int = 20
    head_dim: int = 128
    norm_eps: float = 1e-5
    num_layers: int = 32
    mm_layers: int = 10              # blocks [0, mm_layers) keep separate vid/txt weights
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    window: Tuple[int, int, int] = (4, 3, 3)
    rope_dim: int = 128              # rope_type mmrope3d: 3 axes x (128 // 3 = 42) dims
    expand_ratio: int = 4

    @property
    def mlp_hidden(self) -> int:     # src/models/dit_3b/mlp.py:53-54
        h = int(2 * self.vid_dim * self.expand_ratio / 3)
        return 256 * ((h + 255) // 256)

    @property
    def emb_dim(self) -> int:
        return 6 * self.vid_dim

    @property
    def patch_in_dim(self) -> int:
        t, h, w = self.patch_size
        return self.vid_in_channels * t * h * w

    @property
    def patch_out_dim(self) -> int:
        t, h, w = self.patch_size
        return self.vid_out_channels * t * h * w

    def window_method(self, layer: int) -> str:   # main.yaml:34 alternating regular / shifted
        return "720pwin_by_size_bysize" if layer % 2 == 0 else "720pswin_by_size_bysize"

    def as_dict(self):
        return asdict(self)


DIT_3B = DiTConfig()
# Reduced-width config used by fast parity tests (same code path: 2 separate + 2 shared
# layers, last layer vid-only MLP, regular + shifted windows).
DIT_TINY = DiTConfig(vid_dim=256, heads=2, num_layers=4, mm_layers=2)


@dataclass(frozen=True)
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 16
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    temporal_scale_num: int = 2
    spatial_downsample_factor: int = 8
    temporal_downsample_factor: int = 4
    scaling_factor: float = 0.9152    # configs_3b/main.yaml:60
    shifting_factor: float = 0.0
    slicing_sample_min_size: int = 4  # configs_3b/main.yaml:54 (split_size)


VAE_V3 = VAEConfig()
VAE_TINY = VAEConfig(block_out_channels=(64, 64, 128, 128))
