"""cniic_amd -- MI355X-native (gfx950) implementation of hkapp/cniic's per-pixel compression hot
path: K-means behind `cluster-colors` / `voronoi`, the Huffman-feeding histogram, and the
Hilbert-order gather + neighbour delta behind `delta`.

The product is the C-ABI shared library `libcniic_hip.so` (include/cniic_hip.h, sources under
cniic_amd/csrc/).  This package is the thin host-side mirror of the reference's `Codec` trait
(src/codec.rs:14-19) over that ABI via ctypes; it holds no compute of its own and raises if the
HIP library is missing.
"""
from ._lib import CniicError, Context, lib, lib_path  # noqa: F401
from .codec import AnyCodec, Codec  # noqa: F401

__all__ = ["AnyCodec", "Codec", "Context", "CniicError", "lib", "lib_path"]
