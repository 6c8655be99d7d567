"""Host-side mirror of the reference's Codec trait (src/codec.rs:14-19) and AnyCodec::from_str
(src/codec.rs:41-59) for the hot-path codecs, over the C ABI.

    codec = AnyCodec.from_str("cluster-colors(256)")
    data  = codec.encode(img)          # img: HxWx3 uint8
    img2  = codec.decode(data)         # None where the reference returns None / panics
    codec.name(), codec.is_lossless()
"""
from . import _lib


class Codec:
    """encode / decode / name / is_lossless, like `trait Codec` (src/codec.rs:14-19)."""

    def __init__(self, expr, ctx=None):
        if _lib.codec_parse(expr) is None:
            raise ValueError("Malformed codec argument: %r" % expr)  # main.rs:62-63
        self.expr = expr
        self._ctx = ctx
        self.last_stats = None

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.Context(0)
        return self._ctx

    def encode(self, img, **kw):
        rc, data, st = self.ctx.encode(self.expr, img, **kw)
        self.last_stats = st
        return data

    def decode(self, data):
        rc, img = self.ctx.decode(self.expr, data, allow=(_lib.DECODE,))
        return img if rc == _lib.OK else None

    def name(self):
        return _lib.codec_name(self.expr)

    def is_lossless(self):
        return _lib.codec_is_lossless(self.expr)


class AnyCodec(Codec):
    @classmethod
    def from_str(cls, expr, ctx=None):
        return cls(expr, ctx)
