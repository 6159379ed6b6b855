"""ee_semantic_segmentation_amd - MI355X-native early-exit DeepLabV3 hot path.

Python host code mirroring the reference's module / loss / evaluator surface on
top of libeeseg.so (hand-written HIP for gfx950, C ABI in include/eeseg.h).
"""
__version__ = "0.1.0"
