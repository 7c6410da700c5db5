"""NOVA model assembly: the surface of reference diffnext/models/transformers/transformer_nova.py:29-102.

Architecture tables (`vit_d16w*` conditioning encoders, `vit_d32w*` masked-AR encoders, `mlp_d6w*` / `mlp_d3w1280`
diffusion decoders) and `NOVATransformer3DModel`, whose constructor signature, `config` keys and resulting state_dict
match the reference, so BAAI/nova-* checkpoints load. Geometry rules of the reference: patch = 15 // image_stride + 1,
the conditioning encoder uses twice that patch, RoPE or absolute position embeddings chosen by `rotary_pos_embed`.

Point sets: image_dim = 3 with image_stride = 16 (patch 1) makes every token one (x, y, z) triple.
"""
from ..._compat import ConfigMixin, ModelMixin, register_to_config
from ...utils.registry import Registry
from ..diffusion_mlp import DiffusionMLP
from ..embeddings import LabelEmbed, MaskEmbed, MotionEmbed, PosEmbed, RotaryEmbed3D, TextEmbed, VideoPosEmbed
from ..normalization import AdaLayerNorm
from ..vision_transformer import VisionTransformer
from .transformer_3d import Transformer3DModel

VIDEO_ENCODERS = Registry("video_encoders")
IMAGE_ENCODERS = Registry("image_encoders")
IMAGE_DECODERS = Registry("image_decoders")


def _vit(depth, embed_dim, num_heads, patch_size, image_size, image_dim):
    return VisionTransformer(depth, embed_dim, num_heads, patch_size=patch_size, image_size=image_size, image_dim=image_dim)


def _mlp(depth, embed_dim, patch_size, image_dim, cond_dim):
    return DiffusionMLP(depth, embed_dim, cond_dim, patch_size=patch_size, image_dim=image_dim)


# (model width, attention heads) of the published sizes: 0.3B, 0.6B, 1.4B
for _width, _heads in ((768, 12), (1024, 16), (1536, 16)):
    VIDEO_ENCODERS.register(f"vit_d16w{_width}", _vit, depth=16, embed_dim=_width, num_heads=_heads)
    IMAGE_ENCODERS.register(f"vit_d32w{_width}", _vit, depth=32, embed_dim=_width, num_heads=_heads)
    IMAGE_DECODERS.register(f"mlp_d6w{_width}", _mlp, depth=6, embed_dim=_width)
IMAGE_DECODERS.register("mlp_d3w1280", _mlp, depth=3, embed_dim=1280)


def _latent_geometry(image_size, image_stride):
    """(latent grid, patch of the masked-AR encoder); the conditioning encoder uses twice that patch."""
    if isinstance(image_size, int):
        image_size = (image_size, image_size)
    grid = tuple(side // image_stride for side in image_size)
    return grid, 15 // image_stride + 1


class NOVATransformer3DModel(Transformer3DModel, ModelMixin, ConfigMixin):
    """Text / label conditioned masked-autoregressive generator with a per-token diffusion head."""

    @register_to_config
    def __init__(self, image_dim=None, image_size=None, image_stride=None, text_token_dim=None, text_token_len=None,
                 image_base_size=None, video_base_size=None, video_mixer_rank=None, rotary_pos_embed=False,
                 arch=("", "", "")):
        grid, patch = _latent_geometry(image_size, image_stride)
        video_arch, image_arch, decoder_arch = arch
        video_encoder = VIDEO_ENCODERS.get(video_arch)(image_size=grid, image_dim=image_dim, patch_size=2 * patch)
        image_encoder = IMAGE_ENCODERS.get(image_arch)(image_size=grid, image_dim=image_dim, patch_size=patch)
        width = image_encoder.embed_dim
        image_decoder = IMAGE_DECODERS.get(decoder_arch)(cond_dim=width, image_dim=image_dim, patch_size=patch)

        if rotary_pos_embed:
            video_pos_embed = RotaryEmbed3D(video_encoder.rope.dim, video_base_size[1:])
            image_pos_embed = RotaryEmbed3D(image_encoder.rope.dim, image_base_size)
        else:  # absolute tables: a module in front of the conditioning encoder, a sub-module of the image encoder
            video_pos_embed = VideoPosEmbed(video_encoder.embed_dim, video_base_size)
            image_pos_embed = None
            image_encoder.pos_embed = PosEmbed(width, image_base_size)
        if video_mixer_rank:  # a negative rank selects a plain (no LoRA) AdaLN mixer
            video_encoder.mixer = AdaLayerNorm(video_encoder.embed_dim, max(video_mixer_rank, 0), eps=None)

        # construction order below = parameter-initialisation order under a fixed seed (kept as in the reference)
        mask_embed = MaskEmbed(width)
        text_embed = TextEmbed(text_token_dim, width, text_token_len) if text_token_dim else None
        label_embed = None if text_token_dim else LabelEmbed(width)
        motion_embed = MotionEmbed(video_encoder.embed_dim) if video_base_size[0] > 1 else None
        super().__init__(video_encoder=video_encoder, image_encoder=image_encoder, image_decoder=image_decoder,
                         mask_embed=mask_embed, text_embed=text_embed, label_embed=label_embed,
                         video_pos_embed=video_pos_embed, image_pos_embed=image_pos_embed, motion_embed=motion_embed)
