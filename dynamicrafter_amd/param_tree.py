"""Parameter containers whose state_dict keys equal the reference's checkpoint keys.

The reference's modules own their weights through a deep nn.Module hierarchy (ModuleList / Sequential indices
become key segments such as `input_blocks.4.1.transformer_blocks.0.attn2.to_k_ip.weight`). The HIP path does not
need those modules' forward()s, only their names: `attach_params` grows the same dotted hierarchy out of bare
containers from a {name: shape} table, so `state_dict()` / `load_state_dict(strict=True)` interoperate with
reference checkpoints (including the `temopral_conv` spelling, openaimodel3d.py:190).
"""
import torch
import torch.nn as nn


class ParamNode(nn.Module):
    """A pure container: children and parameters are attached by name."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("ParamNode holds parameters only; compute goes through the HIP plan of the owning model")


def _init_value(name, shape, generator_device):
    """Default init (random, never all-zero): the reference zero-initialises 488-504 tensors, which makes the
    random-init network output exactly 0; synthetic-weight runs want every kernel to do real work."""
    if len(shape) == 0:
        return torch.zeros((), device=generator_device)
    if len(shape) == 1:
        if name.endswith("bias"):
            return torch.randn(shape, device=generator_device) * 0.05
        return 1.0 + 0.1 * torch.randn(shape, device=generator_device)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return torch.randn(shape, device=generator_device) * (0.8 / fan_in ** 0.5)


def attach_params(root, shapes, init=True):
    """Create nn.Parameters under `root` for every dotted name in `shapes` (insertion order preserved)."""
    for name, shape in shapes.items():
        parts = name.split(".")
        node = root
        for seg in parts[:-1]:
            child = node._modules.get(seg)
            if child is None:
                child = ParamNode()
                node.add_module(seg, child)
            node = child
        dev = torch.empty(0).device          # honours `with torch.device(...)`
        val = _init_value(name, tuple(shape), dev) if init else torch.empty(tuple(shape), device=dev)
        node.register_parameter(parts[-1], nn.Parameter(val, requires_grad=False))
    return root


def get_param(root, name):
    node = root
    parts = name.split(".")
    for seg in parts[:-1]:
        node = node._modules[seg]
    return node._parameters[parts[-1]]
