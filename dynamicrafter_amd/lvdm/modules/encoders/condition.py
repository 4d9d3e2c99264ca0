"""Frozen OpenCLIP text / vision towers — OUT OF SCOPE of the HIP hot path.

The reference builds them with open_clip.create_model_and_transforms(..., pretrained="laion2b_s32b_b79k")
(lvdm/modules/encoders/condition.py:188,303): third-party weights that cannot be fetched offline. These
placeholders let the released YAMLs instantiate; callers feed precomputed embeddings (text [B,77,1024],
CLIP-vision [B,257,1280]) to the sampler instead. Calling them raises.
"""
import torch.nn as nn


class _ExternalEncoder(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.init_kwargs = kwargs

    def forward(self, *a, **k):
        raise NotImplementedError(
            f"{type(self).__name__}: OpenCLIP towers are not part of this build (no pretrained weights offline); "
            "pass precomputed conditioning tensors")

    encode = forward


class FrozenOpenCLIPEmbedder(_ExternalEncoder):
    pass


class FrozenOpenCLIPImageEmbedderV2(_ExternalEncoder):
    pass
