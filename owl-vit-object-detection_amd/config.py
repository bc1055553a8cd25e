"""Architecture table for the OWL-ViT vision path.

The reference hard-wires ``google/owlvit-base-patch32`` (reference src/models.py:152-153); the
BASELINE configs ask for B/16 @768 and L/14 @840.  All numbers are the
public HF config values (``OwlViTVisionConfig`` / ``OwlViTTextConfig``).  ``tiny*`` configs exist
only for the parity tests: they keep dh = 64 (what the attention kernels are built for), L = 12 so
the literal ``"layers.11"`` freeze rule (reference src/models.py:175) still selects a layer, and an
L = 14 variant so frozen layers sit *above* the trainable one as in L/14.
"""
from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class OwlConfig:
    name: str
    image_size: int
    patch_size: int
    hidden: int          # D
    heads: int           # H (dh = D / H must be 64)
    mlp: int             # I
    layers: int          # L
    text_dim: int        # Dt (query / class-embedding width)
    n_classes: int = 10  # C; queries = 3 * C (reference src/models.py:155-159)
    ln_eps: float = 1e-5

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def patches(self) -> int:          # P
        return self.grid * self.grid

    @property
    def tokens(self) -> int:           # T = P + 1 (class token first, HF5:338-339)
        return self.patches + 1

    @property
    def tokens_padded(self) -> int:    # Tp: per-image row stride in HBM (multiple of 8)
        return (self.tokens + 7) // 8 * 8

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def queries(self) -> int:          # Q = 3C
        return 3 * self.n_classes

    @property
    def patch_k(self) -> int:          # K of the patch-embed contraction (3 * p * p)
        return 3 * self.patch_size * self.patch_size

    def replace(self, **kw) -> "OwlConfig":
        d = asdict(self)
        d.update(kw)
        return OwlConfig(**d)

    # ---- algorithmic work (2*MACs of matmuls only) -------------------
    def flops_layer(self) -> float:
        T, D, I = self.tokens, self.hidden, self.mlp
        return 8.0 * T * D * D + 4.0 * T * D * I + 4.0 * T * T * D

    def flops_heads(self) -> float:
        P, D, Dt, Q = self.patches, self.hidden, self.text_dim, self.queries
        return 2.0 * P * D * Dt + 2.0 * P * Dt * Q + 2.0 * P * (2 * D * D + 4 * D)

    def flops_forward(self) -> float:
        return self.layers * self.flops_layer() + 2.0 * self.patches * self.patch_k * self.hidden + self.flops_heads()

    def trainable_layer(self) -> int:
        """Index selected by the literal substring rule ``"layers.11" in name``."""
        return 11

    def flops_backward(self) -> float:
        # trainable layer: dX + dW = 2x its forward; frozen layers above it: dX only
        # (linear 1x, attention 2x of their forward parts); heads 2x.
        T, D, I = self.tokens, self.hidden, self.mlp
        lin = 8.0 * T * D * D + 4.0 * T * D * I
        att = 4.0 * T * T * D
        above = self.layers - 1 - self.trainable_layer()
        return 2.0 * self.flops_layer() + above * (lin + 2.0 * att) + 2.0 * self.flops_heads()

    def flops_train_step(self) -> float:
        return self.flops_forward() + self.flops_backward()


CONFIGS = {
    "owlvit-base-patch32": OwlConfig("owlvit-base-patch32", 768, 32, 768, 12, 3072, 12, 512),
    "owlvit-base-patch16": OwlConfig("owlvit-base-patch16", 768, 16, 768, 12, 3072, 12, 512),
    "owlvit-large-patch14": OwlConfig("owlvit-large-patch14", 840, 14, 1024, 16, 4096, 24, 768),
    # parity-test configs (not real checkpoints)
    "tiny": OwlConfig("tiny", 96, 16, 128, 2, 256, 12, 64, n_classes=4),
    "tiny-l14": OwlConfig("tiny-l14", 96, 16, 128, 2, 256, 14, 64, n_classes=4),
    "small": OwlConfig("small", 192, 16, 256, 4, 512, 12, 128, n_classes=10),
}


def get_config(name: str, **overrides) -> OwlConfig:
    name = name.replace("google/", "")
    cfg = CONFIGS[name]
    return cfg.replace(**overrides) if overrides else cfg


@dataclass(frozen=True)
class TextConfig:
    """CLIP-style text tower used once to initialise the query bank (ref src/models.py:155-169; HF ``OwlViTTextConfig``
    defaults for the base models, public ``google/owlvit-large-patch14`` values for L/14).  dh = width / heads = 64."""
    name: str
    width: int
    heads: int
    mlp: int
    layers: int
    proj_dim: int            # = OwlConfig.text_dim (query width); HF needs proj_dim == width (class head out_dim, HF5:1009)
    vocab: int = 49408
    max_pos: int = 16
    ln_eps: float = 1e-5


TEXT_CONFIGS = {
    "owlvit-base-patch32": TextConfig("owlvit-base-text", 512, 8, 2048, 12, 512),
    "owlvit-base-patch16": TextConfig("owlvit-base-text", 512, 8, 2048, 12, 512),
    "owlvit-large-patch14": TextConfig("owlvit-large-text", 768, 12, 3072, 12, 768),
    "tiny": TextConfig("tiny-text", 64, 1, 128, 3, 64, vocab=97, max_pos=16),
    "tiny-l14": TextConfig("tiny-text", 64, 1, 128, 3, 64, vocab=97, max_pos=16),
    "small": TextConfig("small-text", 128, 2, 256, 3, 128, vocab=97, max_pos=16),
}


def get_text_config(name: str) -> TextConfig:
    return TEXT_CONFIGS[name.replace("google/", "")]
