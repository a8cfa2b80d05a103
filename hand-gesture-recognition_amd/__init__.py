"""MI355X-native MultiTaskNet hot path (GELAN backbone + ViT encoder + pose/cls heads).

Drop-in for ``model/multitasknet.py`` of yingkunwu/hand-gesture-recognition: same
``MultiTaskNet(num_joints, num_classes, image_size)`` constructor, ``forward(x) ->
(cls_out, hmap_out, attnmap)`` and ``state_dict`` layout; all arithmetic runs in hand-written
HIP kernels (``csrc/``) reached through the C ABI of ``include/mtn.h``.
"""
__version__ = "0.1.0"
