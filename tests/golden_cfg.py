"""Small-network hyper-parameters shared by tests/golden/make_golden.py's fixtures and the tests that replay them
(real topology of the released UNet / AE, narrow widths; widths are multiples of 64 = one MFMA K slice)."""
TINY_UNET = dict(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                 channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64, transformer_depth=1, context_dim=128,
                 use_linear=True, use_checkpoint=False, temporal_conv=True, temporal_attention=True,
                 temporal_selfatt_only=True, use_relative_position=False, use_causal_attention=False,
                 temporal_length=4, addition_attention=True, image_cross_attention=True, default_fs=10,
                 fs_condition=True)
TINY_AE = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
               num_res_blocks=2, attn_resolutions=[], dropout=0.0)


# ---- stand-ins for the two frozen OpenCLIP towers (their weights are not available offline): deterministic torch
# modules with the towers' interfaces, used IDENTICALLY by the fixture generator (inside the reference model) and by
# the tests (inside this package's model). Test fixtures, not product code.
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

TINY_RESAMPLER = dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=64, output_dim=128,
                      ff_mult=4, video_length=4)


class ToyImageEmbedder(nn.Module):
    """image [B,3,H,W] -> tokens [B, 9, 64] (the vision tower returns [B, 257, 1280]): 3x3 average-pooled colours
    through a fixed linear map."""

    def __init__(self, seed=15):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("w", torch.empty(3, 64).normal_(generator=g))
        self.register_buffer("pos", torch.empty(9, 64).normal_(generator=g) * 0.1)

    def forward(self, image):
        p = torch.nn.functional.adaptive_avg_pool2d(image.float(), 3)            # [B,3,3,3]
        tok = p.flatten(2).transpose(1, 2)                                        # [B,9,3]
        return tok @ self.w + self.pos


class ToyTextEmbedder(nn.Module):
    """list of prompts -> [B, 77, 128] (the text tower returns [B, 77, 1024]); content seeded by the prompt text."""

    def __init__(self, dim=128):
        super().__init__()
        self.dim = dim
        self.register_buffer("_dev", torch.zeros(1))

    def encode(self, prompts):
        out = []
        for s in prompts:
            g = torch.Generator().manual_seed(1000 + sum(ord(ch) for ch in s))
            out.append(torch.empty(77, self.dim).normal_(generator=g))     # not torch.randn: tests patch that name
        return torch.stack(out).to(self._dev.device)

    forward = encode
