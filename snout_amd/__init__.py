"""snout_amd — MI355X-native IQ -> packets receive path behind Snout's scan interface.

The compute path is libsnout_rx.so (hand-written HIP for gfx950, C ABI in include/snout_rx.h);
this package is the thin host side: ctypes binding (:mod:`snout_amd._ffi`, :mod:`snout_amd.rx`),
scan handlers mirroring the reference's (:mod:`snout_amd.scan`), signal generators
(:mod:`snout_amd.synth`) and multi-GPU sharding (:mod:`snout_amd.dist`).
"""
__version__ = "0.1.0"
