"""GELAN backbone -- host-side mirror of the reference's ``model/gelan.py`` interface.

These modules own the parameters (same names, shapes, construction order and default
initialisation as the reference, so ``state_dict`` keys and seeded init match) and describe the
layer graph; the arithmetic runs in HIP kernels scheduled by ``engine.Engine``.  None of the
torch modules held here is ever *called*.

Reference: ``autopad`` gelan.py:5-15, ``Conv`` :18-56, ``ResBasicBlock`` :59-87,
``GELANBlock`` :124-142, ``GELANNet`` :145-176 (both spec rows, and optionally the fourth
stage the reference keeps commented out).  ``ResBottleneck`` (:90-121) is never instantiated by
the reference's spec table (:148-151) and is not provided.
"""
import torch
from torch import nn


def autopad(k, p=None, d=1):
    """'same' padding for kernel ``k`` (gelan.py:5-15): k//2 at dilation 1."""
    if d > 1:
        k = d * (k - 1) + 1
    return k // 2 if p is None else p


class Conv(nn.Module):
    """Conv2d(bias=False) + BatchNorm2d + SiLU (``act=True``) or identity (gelan.py:18-56)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if g != 1 or d != 1:
            raise NotImplementedError("the MultiTaskNet hot path only uses groups=1, dilation=1")
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.c1, self.c2, self.k, self.s, self.has_act = c1, c2, k, s, act is True

    def forward(self, x):
        raise RuntimeError("Conv runs inside the fused plan of GELANNet / MultiTaskNet (engine.py)")


class ResBasicBlock(nn.Module):
    """SiLU(x + bn(conv3x3(SiLU(bn(conv3x3(x)))))) with c1 == c2 (gelan.py:59-87)."""

    def __init__(self, c1, c2, shortcut=True):
        super().__init__()
        if c1 != c2 or not shortcut:
            raise NotImplementedError("GELANNet only builds ResBasicBlock(c, c, shortcut=True)")
        self.cv1 = Conv(c1, c2, 3, 1)
        self.cv2 = Conv(c2, c2, 3, 1, act=False)

    def forward(self, x):
        raise RuntimeError("ResBasicBlock runs inside the fused plan (engine.py)")


class GELANBlock(nn.Module):
    """cv1 -> chunk(2) -> cv2 -> cv3 -> cat -> cv4 (gelan.py:124-142); ``nblocks`` residual blocks per
    branch (1 in the 'small' spec, 2 in 'large')."""

    def __init__(self, c_in, c_out, c_hid1, c_hid2, block=ResBasicBlock, nblocks=1):
        super().__init__()
        if nblocks < 1 or c_hid1 // 2 != c_hid2:
            raise NotImplementedError("GELANBlock needs c_hid1 // 2 == c_hid2 (all reference specs) and nblocks >= 1")
        self.cv1 = Conv(c_in, c_hid1, 1, 1)
        self.cv2 = nn.Sequential(*([block(c_hid1 // 2, c_hid2)] + [block(c_hid2, c_hid2) for _ in range(nblocks - 1)]))
        self.cv3 = nn.Sequential(*(block(c_hid2, c_hid2) for _ in range(nblocks)))
        self.cv4 = Conv(c_hid1 + 2 * c_hid2, c_out, 1, 1)
        self.c_in, self.c_out, self.c_hid1, self.c_hid2, self.nblocks = c_in, c_out, c_hid1, c_hid2, nblocks

    def forward(self, x):
        raise RuntimeError("GELANBlock runs inside the fused plan (engine.py)")


class GELANNet(nn.Module):
    """GELANNet (gelan.py:145-176): 'small' = 1 block per branch, 'large' = 2 (spec table :148-151).

    ``stages=3`` is what the reference builds (total stride 16, 3 -> 512 channels).  ``stages=4``
    adds the ``down3`` / ``cspelan4`` pair the reference keeps commented out (:162-163, :174-175):
    total stride 32, still 512 channels.
    """

    SPEC = {"small": [1, 1, 1, 1], "large": [2, 2, 2, 2]}

    def __init__(self, gelan_type="small", stages=3):
        super().__init__()
        if gelan_type not in self.SPEC or stages not in (3, 4):
            raise ValueError(f"GELANNet: unknown spec {gelan_type!r} / stages={stages}")
        layers = self.SPEC[gelan_type]
        self.conv1 = Conv(3, 64, 3, 2)
        self.conv2 = Conv(64, 128, 3, 2)
        self.cspelan1 = GELANBlock(128, 128, 128, 64, ResBasicBlock, layers[0])
        self.down1 = Conv(128, 256, 3, 2)
        self.cspelan2 = GELANBlock(256, 256, 256, 128, ResBasicBlock, layers[1])
        self.down2 = Conv(256, 512, 3, 2)
        self.cspelan3 = GELANBlock(512, 512, 512, 256, ResBasicBlock, layers[2])
        if stages == 4:
            self.down3 = Conv(512, 512, 3, 2)
            self.cspelan4 = GELANBlock(512, 512, 512, 256, ResBasicBlock, layers[3])
        self.gelan_type, self.stages, self.total_stride = gelan_type, stages, 4 << (stages - 1)

    def stage_modules(self):
        """[(GELANBlock, following down Conv or None)] in forward order"""
        blocks = [getattr(self, f"cspelan{i}") for i in range(1, self.stages + 1)]
        downs = [getattr(self, f"down{i}") for i in range(1, self.stages)] + [None]
        return list(zip(blocks, downs))

    def forward(self, x):
        """gelan.py:165-176.  The backbone runs in the plan of the MultiTaskNet that owns this module (``model.encoder(x)``); a
        GELANNet constructed on its own has parameters but no plan."""
        owner = self.__dict__.get("_owner")
        if owner is None:
            raise RuntimeError("a GELANNet constructed on its own has no execution plan: use it as MultiTaskNet(...).encoder, "
                               "or engine.Engine.backbone_* directly")
        return owner.encoder_forward(x)
