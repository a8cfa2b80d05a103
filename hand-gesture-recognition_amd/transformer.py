"""ViT encoder + heads -- host-side mirror of the reference's ``model/transformer.py`` interface.

Parameter containers with the reference's names / shapes / construction order; the arithmetic
runs in HIP kernels scheduled by ``engine.Engine``.

Reference: ``pos_emb_sincos_2d`` transformer.py:9-26, ``FeedForward`` :29-42, ``Attention``
:45-77, ``Transformer`` :80-96, ``ViT`` :99-152.
"""
import torch
from torch import nn


def pos_emb_sincos_2d(h, w, dim, temperature=10000, dtype=torch.float32):
    """2-D sin/cos table [h*w, dim], computed once on the host (transformer.py:9-26).

    Faithful to the reference's quirk: omega_k = 1 / temperature**k for k = 0..dim/4-1 with no
    normalisation of k, so temperature**k overflows fp32 to inf for k >= 10 and those columns are
    sin = 0 / cos = 1.  Column blocks are (sin x, cos x, sin y, cos y); x = column index.
    """
    assert dim % 4 == 0, "dimension must be divisible by 4"
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    omega = 1.0 / (temperature ** torch.arange(dim // 4, dtype=dtype))
    ys = ys.flatten()[:, None] * omega[None, :]
    xs = xs.flatten()[:, None] * omega[None, :]
    return torch.cat((xs.sin(), xs.cos(), ys.sin(), ys.cos()), dim=1).type(dtype)


class FeedForward(nn.Module):
    """LayerNorm -> Linear -> GELU(erf) -> Linear, dropout p = 0 (transformer.py:29-42)."""

    def __init__(self, dim, hidden_dim, dropout=0.0):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("the reference always uses dropout = 0")
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), nn.Identity(),
                                 nn.Identity(), nn.Linear(hidden_dim, dim), nn.Identity())

    def forward(self, x):
        raise RuntimeError("FeedForward runs inside the fused plan (engine.py)")


class Attention(nn.Module):
    """Pre-LN multi-head attention with bias-free qkv/out projections (transformer.py:45-77)."""

    def __init__(self, dim, heads, head_dim):
        super().__init__()
        inner = head_dim * heads
        self.heads, self.head_dim, self.scale = heads, head_dim, head_dim ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def forward(self, x):
        raise RuntimeError("Attention runs inside the fused plan (engine.py)")


class Transformer(nn.Module):
    """depth x [x = attn(x) + x ; x = ff(x) + x], returns the last attention map (:80-96)."""

    def __init__(self, dim, depth, heads, head_dim, mlp_dim, dropout=0.0):
        super().__init__()
        self.layers = nn.ModuleList(
            nn.ModuleList([Attention(dim, heads, head_dim), FeedForward(dim, mlp_dim, dropout)])
            for _ in range(depth))

    def forward(self, x):
        raise RuntimeError("Transformer runs inside the fused plan (engine.py)")


class ViT(nn.Module):
    """Token assembly + transformer + classification head + pose head (transformer.py:99-152)."""

    def __init__(self, num_classes, num_joints, feature_size, dim, depth, heads, head_dim, mlp_dim,
                 dropout=0.0):
        super().__init__()
        # plain attribute like the reference (:103-107): not a Parameter, not in state_dict
        self.pos_embedding = pos_emb_sincos_2d(h=feature_size[0], w=feature_size[1], dim=dim)
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.transformer = Transformer(dim, depth, heads, head_dim, mlp_dim, dropout)
        self.mlp_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))
        # index 0 is the reference's parameter-free ReLU(inplace=True)
        self.simple_decoder = nn.Sequential(nn.Identity(), nn.Conv2d(dim, num_joints, 1, 1, 0))
        self.feature_size = list(feature_size)

    def forward(self, x):
        """transformer.py:129-152 on projected features [B, dim, h, w].  Runs in the plan of the MultiTaskNet that owns this
        module (``model.decoder(feats)``); the reference adds the positional embedding INTO its argument (``x += pe`` on a view
        of it, :135) -- here the argument is left as it was."""
        owner = self.__dict__.get("_owner")
        if owner is None:
            raise RuntimeError("a ViT constructed on its own has no execution plan: use it as MultiTaskNet(...).decoder")
        return owner.decoder_forward(x)
