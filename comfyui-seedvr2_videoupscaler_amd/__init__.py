"""MI355X-native SeedVR2 hot path (NaDiT forward + causal-Conv3d video VAE) behind the
reference runner / node API.  See DESIGN.md.  Import is side-effect free; the HIP C-ABI
library is loaded on first use and its absence is a hard error (no CPU fallback)."""
__version__ = "0.1.0"
