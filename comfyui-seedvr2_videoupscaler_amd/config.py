"""Model hyper-parameters of the SeedVR2 hot path, transcribed from the reference YAMLs
(omegaconf is not available here; values cited per key).

  configs_3b/main.yaml:11-36            -> DIT_3B
  configs_7b/main.yaml:11-33            -> DIT_7B
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
    # ---- the 7B family (configs_7b/main.yaml:11-33, src/models/dit_7b) differs in exactly these:
    mlp_type: str = "swiglu"         # "normal": Linear+bias -> GELU(tanh) -> Linear+bias, hidden = 4 d (dit_7b/mlp.py:28-43)
    rope_type: str = "mmrope3d"      # "rope3d": video tokens only, "pixel" frequencies linspace(1, 128, 10) * pi,
                                     # positions linspace(-1, 1, n) along each window axis, 3 x 20 dims (dit_7b/rope.py)
    out_norm: bool = True            # 3B: vid_out_norm + vid_out_ada before the output projection; 7B: none
    last_vid_only: bool = True       # 3B: the last block's ada / mlp are video-only (mmsr_block.py:73-81); 7B: full block

    @property
    def mlp_hidden(self) -> int:     # src/models/dit_3b/mlp.py:53-54 / dit_7b/mlp.py:35
        if self.mlp_type == "normal":
            return self.vid_dim * self.expand_ratio
        h = int(2 * self.vid_dim * self.expand_ratio / 3)
        return 256 * ((h + 255) // 256)

    @property
    def rope_freqs(self) -> int:     # frequencies per axis
        return (self.head_dim // 2 // 3) // 2 if self.rope_type == "rope3d" else (self.rope_dim // 3) // 2

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
DIT_7B = DiTConfig(vid_dim=3072, heads=24, num_layers=36, mm_layers=36, mlp_type="normal", rope_type="rope3d",
                   out_norm=False, last_vid_only=False)
DIT_7B_TINY = DiTConfig(vid_dim=256, heads=2, num_layers=4, mm_layers=4, mlp_type="normal", rope_type="rope3d",
                        out_norm=False, last_vid_only=False)
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
