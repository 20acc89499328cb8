"""Parameter containers shared by the resampler and the gated cross-attention block, plus small helpers.

`FeedForward` keeps the reference's Sequential structure (flamingo_mini/utils.py:22-50) so state_dict keys stay
`0.weight, 0.bias, 1.weight, 3.weight`; the arithmetic itself runs inside the fused HIP kernels
(LayerNorm -> GEMM+activation epilogue -> GEMM+residual epilogue), never through these modules' forward().
"""
from __future__ import annotations

import torch
from torch import nn

ACTIVATIONS = ("gelu", "sqrelu", "relu")


class SquaredReLU(nn.Module):
    """relu(x)**2 (the activation the Flamingo paper used)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.relu(x).square()


def _activation_module(act: str) -> nn.Module:
    return {"gelu": nn.GELU, "sqrelu": SquaredReLU, "relu": nn.ReLU}[act]()


def FeedForward(dim: int, mult: int = 4, act: str = "gelu") -> nn.Sequential:
    assert act in ACTIVATIONS, f"act. can only be one of {ACTIVATIONS}"
    hidden = int(dim * mult)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden, bias=False), _activation_module(act),
                         nn.Linear(hidden, dim, bias=False))


def feedforward_params(ffw: nn.Sequential):
    return [ffw[0].weight, ffw[0].bias, ffw[1].weight, ffw[3].weight]


def get_common_prefix_length(x: torch.Tensor) -> int:
    """Length of the prefix shared by all rows of a 2-D id matrix (used by score_sequences)."""
    same = (x[:1] == x).all(dim=0)
    differing = (~same).nonzero()
    return int(differing[0]) if differing.numel() else x.size(1)


def unzip(pairs):
    return list(zip(*pairs))


def load_image(path: str):
    from PIL import Image
    return Image.open(path)


def load_url(url: str):
    import requests
    from PIL import Image
    return Image.open(requests.get(url, stream=True).raw)
