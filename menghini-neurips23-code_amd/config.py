"""CLIP tower dimensions for the encoders the reference's scripts name
(VIS_ENCODER env var, scripts/run_pseudolabels_ssl.sh:4; BASELINE.json configs),
plus tiny shapes used by tests and golden fixtures."""
from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class ClipDims:
    name: str
    embed_dim: int
    image_resolution: int
    vision_layers: int
    vision_width: int
    vision_patch_size: int
    context_length: int
    vocab_size: int
    transformer_width: int
    transformer_heads: int
    transformer_layers: int

    @property
    def vision_heads(self) -> int:
        return self.vision_width // 64

    @property
    def grid(self) -> int:
        return self.image_resolution // self.vision_patch_size

    @property
    def vision_seq(self) -> int:
        return self.grid * self.grid + 1

    def to_dict(self):
        return asdict(self)


_D = ClipDims
CLIP_CONFIGS = {
    "ViT-B/32": _D("ViT-B/32", 512, 224, 12, 768, 32, 77, 49408, 512, 8, 12),
    "ViT-B/16": _D("ViT-B/16", 512, 224, 12, 768, 16, 77, 49408, 512, 8, 12),
    "ViT-L/14": _D("ViT-L/14", 768, 224, 24, 1024, 14, 77, 49408, 768, 12, 12),
    "ViT-L/14@336px": _D("ViT-L/14@336px", 768, 336, 24, 1024, 14, 77, 49408, 768, 12, 12),
    # test-only shapes: head dim is always 64 (as in every CLIP ViT), widths are
    # multiples of 128 so the MFMA GEMM tiles apply unchanged.
    "tiny": _D("tiny", 128, 32, 2, 128, 8, 77, 49408, 128, 2, 2),
    # ViT-L/14@336px geometry (patch 14 -> K = 588 padded to 640, S = 577) at test width
    "tinyL336": _D("tinyL336", 128, 336, 2, 128, 14, 77, 49408, 128, 2, 2),
    "small": _D("small", 256, 64, 3, 256, 16, 77, 49408, 256, 4, 2),
}

SOT_TOKEN = 49406
EOT_TOKEN = 49407
X_TOKEN = 343  # BPE id of "x</w>", the placeholder CustomTextEncoder writes (models/clip_encoders.py:54-57)


def get_dims(name: str) -> ClipDims:
    if name not in CLIP_CONFIGS:
        raise RuntimeError(f"Model {name} not found; available models = {list(CLIP_CONFIGS)}")
    return CLIP_CONFIGS[name]
