"""Codec::decode with the stream (and the image) resident in HBM: the payload is decoded where it lies -- at whatever byte
alignment the serialised decoder leaves it -- and only the head of the stream (dimensions + decoder) visits the host.  Same
pixels as the host-buffer call and as the oracle; the context options that replace the route environment variables."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import cniic_amd
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    yield ctx, torch, dev
    ctx.close()


@pytest.mark.parametrize("expr", ["hufman", "delta", "cluster-colors(8)", "ccol(200)", "hilbert(rle)", "voronoi(12)"])
@pytest.mark.parametrize("shape", [(3, 5), (64, 64), (97, 131), (256, 320)])
def test_device_resident_decode_equals_oracle(env, expr, shape):
    ctx, torch, dev = env
    from cniic_amd import _lib, synth
    h, w = shape
    img = synth.photo(w, h, synth.SEED0 + 77 + h)
    rc, data, _ = ctx.encode(expr, img, allow=(_lib.TOO_FEW_POINTS, _lib.FEW_ACTIVE))
    if rc != 0:
        pytest.skip("fewer colours than clusters")
    rco, exp = O.decode(expr, data)
    assert rco == 0
    ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, 0)          # the parallel decoder whatever the size
    try:
        for shift in (0, 1, 2, 3):                  # the stream at every byte alignment (the payload's own varies with the decoder's size)
            buf = torch.zeros(len(data) + 64, dtype=torch.uint8, device=dev)
            buf[shift:shift + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
            out = torch.zeros(h * w * 3 + 16, dtype=torch.uint8, device=dev)
            rc, dw, dh = ctx.decode_into(expr, buf[shift:], len(data), out)
            assert rc == 0 and (dw, dh) == (w, h)
            assert np.array_equal(out[:h * w * 3].cpu().numpy().reshape(h, w, 3), exp), (expr, shape, shift)
            assert not out[h * w * 3:].any()
        # a host stream into a device image, and a device stream into a host image
        out = torch.zeros(h * w * 3, dtype=torch.uint8, device=dev)
        rc, _, _ = ctx.decode_into(expr, np.frombuffer(data, np.uint8), len(data), out)
        assert rc == 0 and np.array_equal(out.cpu().numpy().reshape(h, w, 3), exp)
        hout = np.zeros(h * w * 3, np.uint8)
        rc, _, _ = ctx.decode_into(expr, buf[3:], len(data), hout)
        assert rc == 0 and np.array_equal(hout.reshape(h, w, 3), exp)
    finally:
        ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, None)


def test_device_resident_truncated_streams_fail_like_the_oracle(env):
    ctx, torch, dev = env
    from cniic_amd import _lib, synth
    img = synth.photo(120, 90, synth.SEED0 + 5)
    ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, 0)
    try:
        for expr in ("hufman", "delta", "cluster-colors(16)"):
            rc, data, _ = ctx.encode(expr, img)
            full = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
            out = torch.zeros(120 * 90 * 3, dtype=torch.uint8, device=dev)
            for cut in (1, 5, 200, len(data) // 2, len(data) - 9, len(data) - 8):
                rc, _, _ = ctx.decode_into(expr, full, len(data) - cut, out, allow=(_lib.DECODE,))
                rco, _ = O.decode(expr, data[:len(data) - cut])
                assert (rc == 0) == (rco == 0), (expr, cut)
    finally:
        ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, None)


def test_decoder_longer_than_the_first_head_fetch(env):
    """a decoder of more than 64 KiB (the first piece of a device-resident stream the host looks at): the head is fetched further"""
    ctx, torch, dev = env
    from cniic_amd import _lib, synth
    img = synth.uniform(300, 200, synth.SEED0 + 9)                     # ~60000 distinct colours: 12 bytes each in the decoder
    rc, data, _ = ctx.encode("hufman", img)
    assert len(data) > 300 * 200 * 9
    out = torch.zeros(300 * 200 * 3, dtype=torch.uint8, device=dev)
    rc, dw, dh = ctx.decode_into("hufman", torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev), len(data), out)
    assert rc == 0 and np.array_equal(out.cpu().numpy().reshape(200, 300, 3), img)


@pytest.mark.parametrize("second", [None, "1", "100000000"])
@pytest.mark.parametrize("shape,kind", [((200, 300), "photo"), ((256, 256), "uniform"), ((700, 500), "photo")])
def test_delta_decoder_longer_than_the_first_look(env, monkeypatch, shape, kind, second):
    """a `delta` decoder that outgrows the head of the stream the host looks at first: a second, longer look (512 KiB) parses it on
    the host; one that outgrows that too -- or CNIIC_TRIE_HOST_SECOND=1: no second look -- is parsed on the GPU.  Same pixels."""
    ctx, torch, dev = env
    from cniic_amd import _lib, synth
    h, w = shape
    img = getattr(synth, kind)(w, h, synth.SEED0 + 3 + h)
    rc, data, _ = ctx.encode("delta", img)
    assert rc == 0 and len(data) > 8192 * 2
    if second is not None:
        monkeypatch.setenv("CNIIC_TRIE_HOST_SECOND", second)
    ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, 0)
    try:
        out = torch.zeros(h * w * 3, dtype=torch.uint8, device=dev)
        full = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        torch.cuda.synchronize()
        rc, dw, dh = ctx.decode_into("delta", full, len(data), out)
        assert rc == 0 and (dw, dh) == (w, h) and np.array_equal(out.cpu().numpy().reshape(h, w, 3), img)
        rc, back = ctx.decode("delta", data)     # the stream in host memory
        assert rc == 0 and np.array_equal(back, img)
        for cut in (9, len(data) // 3, len(data) // 2):
            rc, _, _ = ctx.decode_into("delta", full, len(data) - cut, out, allow=(_lib.DECODE,))
            assert (rc == 0) == (O.decode("delta", data[:len(data) - cut])[0] == 0), cut
    finally:
        ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, None)


def test_context_options_replace_the_environment(env, monkeypatch):
    ctx, torch, dev = env
    from cniic_amd import _lib, synth
    monkeypatch.delenv("CNIIC_SP_MIN_PIXELS", raising=False)
    assert ctx.get_opt(_lib.OPT_SP_MIN_PIXELS) == 1 << 20
    monkeypatch.setenv("CNIIC_SP_MIN_PIXELS", "123")
    assert ctx.get_opt(_lib.OPT_SP_MIN_PIXELS) == 123                 # unset option: the environment, read per call
    ctx.set_opt(_lib.OPT_SP_MIN_PIXELS, 0)
    assert ctx.get_opt(_lib.OPT_SP_MIN_PIXELS) == 0                   # a set option wins
    img = synth.photo(96, 64, synth.SEED0 + 1)
    rc, a, _ = ctx.encode("cluster-colors(16)", img)                   # through the pixel partition
    ctx.set_opt(_lib.OPT_SP_MIN_PIXELS, 1 << 40)
    rc, b, _ = ctx.encode("cluster-colors(16)", img)                   # through the dense table
    ctx.set_opt(_lib.OPT_SP_MIN_PIXELS, None)
    assert ctx.get_opt(_lib.OPT_SP_MIN_PIXELS) == 123
    rco, e, _ = O.encode("cluster-colors(16)", img, mode=O.MODE_L)
    assert a == b == e
    ctx.set_opt(_lib.OPT_DELTA_ROUTE, 32)
    rc, d32, _ = ctx.encode("delta", img)
    ctx.set_opt(_lib.OPT_DELTA_ROUTE, None)
    rc, d16, _ = ctx.encode("delta", img)
    assert d32 == d16 == O.encode("delta", img)[1]
    with pytest.raises(Exception):
        ctx.set_opt(99, 1)


@pytest.mark.parametrize("expr", ["hufman", "delta", "cluster-colors(8)", "ccol(200)"])
@pytest.mark.parametrize("shape,kind", [((1, 1), "photo"), ((5, 7), "photo"), ((64, 64), "photo"), ((130, 97), "photo"), ((200, 300), "uniform"), ((512, 512), "uniform")])
def test_decoder_parsed_on_the_gpu_equals_oracle(env, monkeypatch, expr, shape, kind):
    """Dec::deserialize on the GPU (k_trieparse.hip: chunk maps -> composition -> node records -> right children -> leaf codes),
    forced for every decoder whatever its size: same pixels as the oracle; host and device streams; every cut of the stream fails
    or succeeds as the oracle's decoder does"""
    ctx, torch, dev = env
    from cniic_amd import _lib, synth
    h, w = shape
    img = getattr(synth, kind)(w, h, synth.SEED0 + 31 + h)
    rc, data, _ = ctx.encode(expr, img, allow=(_lib.TOO_FEW_POINTS, _lib.FEW_ACTIVE))
    if rc != 0:
        pytest.skip("fewer colours than clusters")
    rco, exp = O.decode(expr, data)
    assert rco == 0
    monkeypatch.setenv("CNIIC_TEST_TRIE_GPU", "1")
    ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, 0)
    try:
        rc, back = ctx.decode(expr, data)
        assert rc == 0 and np.array_equal(back, exp)
        full = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        out = torch.zeros(h * w * 3, dtype=torch.uint8, device=dev)
        rc, _, _ = ctx.decode_into(expr, full, len(data), out)
        assert rc == 0 and np.array_equal(out.cpu().numpy().reshape(h, w, 3), exp)
        rng = np.random.default_rng(h * w)
        cuts = [1, 2, 9, len(data) // 3, len(data) - 9, len(data) - 12] + rng.integers(1, max(2, len(data) - 8), 6).tolist()
        for cut in cuts:
            if not 0 < cut < len(data) - 7:
                continue
            rc, _, _ = ctx.decode_into(expr, full, len(data) - cut, out, allow=(_lib.DECODE,))
            rco, _ = O.decode(expr, data[:len(data) - cut])
            assert (rc == 0) == (rco == 0), (expr, shape, cut)
        # a corrupted tag / symbol inside the decoder: both refuse, or both decode the same pixels
        for at in rng.integers(8, min(len(data), 8 + 2000), 8).tolist():
            bad = bytearray(data)
            bad[at] ^= 0x5A
            rc, b2 = ctx.decode(expr, bytes(bad), allow=(_lib.DECODE,))
            rco, e2 = O.decode(expr, bytes(bad))
            assert (rc == 0) == (rco == 0), (expr, shape, at)
            if rc == 0:
                assert np.array_equal(b2, e2)
    finally:
        ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, None)


@pytest.mark.parametrize("phases", ["1", "0", None, "verdict-off", "verdict-eager"])
@pytest.mark.parametrize("expr,kind,shape", [("delta", "U", (512, 512)), ("hufman", "U", (256, 256)), ("delta", "P", (300, 200)),
                                              ("cluster-colors(16)", "U", (128, 128)), ("hufman", "P", (97, 131)), ("hufman", "P", (640, 512))])
def test_streams_that_do_not_fall_into_step(env, monkeypatch, expr, kind, shape, phases):
    """A code whose words are nearly all the same length (uniform noise over a small alphabet: `delta` on the 512 x 512 U image) does not
    self-synchronise: such a stream is decoded from every possible entry phase of every subsequence and the phase maps are composed
    (k_hd_phase_maps / k_hd_phase_chain).  CNIIC_HD_PHASES=1 forces that route for every stream, 0 forbids it (the checks go on one
    subsequence at a time), unset: taken when the blind checks have not settled a short stream -- or at once when pass 0 itself finds
    that its blocks' out-of-step lists do not shrink (the verdict; CNIIC_HD_HOPELESS_PCT=0 switches it off, 1 makes every block
    with a list of 64 give the stream up).  The same pixels every way, at every byte alignment of the stream."""
    ctx, torch, dev = env
    from cniic_amd import _lib, synth
    h, w = shape
    img = (synth.uniform if kind == "U" else synth.photo)(w, h, synth.SEED0 + 5 + h)
    rc, data, _ = ctx.encode(expr, img, allow=(_lib.TOO_FEW_POINTS, _lib.FEW_ACTIVE))
    assert rc == 0
    rco, exp = O.decode(expr, data)
    assert rco == 0
    if phases in ("verdict-off", "verdict-eager"):
        monkeypatch.setenv("CNIIC_HD_HOPELESS_PCT", "0" if phases == "verdict-off" else "1")
    elif phases is not None:
        monkeypatch.setenv("CNIIC_HD_PHASES", phases)
    ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, 0)
    try:
        for shift in (0, 3):
            buf = torch.zeros(len(data) + 64, dtype=torch.uint8, device=dev)
            buf[shift:shift + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
            out = torch.zeros(h * w * 3 + 16, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            rc, dw, dh = ctx.decode_into(expr, buf[shift:], len(data), out)
            assert rc == 0 and (dw, dh) == (w, h)
            assert np.array_equal(out[:h * w * 3].cpu().numpy().reshape(h, w, 3), exp)
    finally:
        ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, None)


@pytest.mark.parametrize("shape", [(5, 5), (7, 9), (64, 65)])
@pytest.mark.parametrize("zeros", [1, 4098, 9000])
def test_rle_zero_count_records_behind_the_image_are_never_walked(env, shape, zeros):
    """a hilbert(rle) stream whose records fill the image, followed by thousands of ZERO-count records (they start
    at colour n, are never read by the decoder -- hilbertc.rs:304-337 zipped with the scan -- and so are legal) and one long run.  The
    thread that straddles n (w*h is no multiple of 16) used to walk over them without bound.  Same pixels as the oracle, no hang."""
    ctx, torch, dev = env
    from cniic_amd import synth
    h, w = shape
    img = synth.uniform(w, h, synth.SEED0 + 13 + zeros)          # noise: every run has length 1
    rc, data, _ = ctx.encode("hilbert(rle)", img)
    assert rc == 0
    rec0 = bytes([0, 3, 0, 0, 0, 0, 0, 0, 0, 9, 9, 9])           # count 0, colour (9, 9, 9): well-formed bytes, count the decoder would refuse IF it read it
    last = bytes([255, 3, 0, 0, 0, 0, 0, 0, 0, 1, 2, 3])
    hostile = data + rec0 * zeros + last
    rco, exp = O.decode("hilbert(rle)", hostile)
    assert rco == 0 and np.array_equal(exp, img)
    rc, back = ctx.decode("hilbert(rle)", hostile)
    assert rc == 0 and np.array_equal(back, img)
    full = torch.frombuffer(bytearray(hostile), dtype=torch.uint8).to(dev)
    out = torch.zeros(h * w * 3, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    rc, _, _ = ctx.decode_into("hilbert(rle)", full, len(hostile), out)
    assert rc == 0 and np.array_equal(out.cpu().numpy().reshape(h, w, 3), img)
    # and a zero count INSIDE the image is still the reference's assert
    bad = bytearray(hostile)
    bad[8 + 12 * (h * w // 2)] = 0
    assert O.decode("hilbert(rle)", bytes(bad))[0] != 0
    rc, _ = ctx.decode("hilbert(rle)", bytes(bad), allow=(-6,))
    assert rc != 0


@pytest.mark.parametrize("keep", ["0", "1"])
@pytest.mark.parametrize("lut3", ["0", "1"])
@pytest.mark.parametrize("lut1", ["0", "1"])
@pytest.mark.parametrize("expr,shape", [("delta", (300, 420)), ("hufman", (256, 320)), ("cluster-colors(200)", (240, 320)), ("hilbert(rle)", (128, 200))])
def test_decoder_routes_round_5(env, monkeypatch, expr, shape, keep, lut3, lut1):
    """the parallel decoder with and without its kept symbols (CNIIC_HD_KEEP: the passes' symbols copied to their places by k_hd_compact,
    subsequences of more than 64 symbols decoded again -- the 8-bit codes of `cluster-colors` are mostly those) and with and without the
    third tables (CNIIC_HD_LUT3: codes beyond the second table's bits resolved by one more read), with the first table in LDS or
    without it (CNIIC_HD_LUT1: long-coded streams go straight to the second table and the blocks take half the LDS): the oracle's
    pixels on every route"""
    ctx, torch, dev = env
    from cniic_amd import _lib, synth
    monkeypatch.setenv("CNIIC_HD_KEEP", keep)
    monkeypatch.setenv("CNIIC_HD_LUT3", lut3)
    monkeypatch.setenv("CNIIC_HD_LUT1", lut1)
    h, w = shape
    img = synth.photo(w, h, synth.SEED0 + 31 + h)
    rc, data, _ = ctx.encode(expr, img)
    rco, exp = O.decode(expr, data)
    assert rc == 0 and rco == 0
    ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, 0)
    try:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        out = torch.zeros(h * w * 3 + 16, dtype=torch.uint8, device=dev)
        rc, dw, dh = ctx.decode_into(expr, buf, len(data), out)
        assert rc == 0 and (dw, dh) == (w, h)
        assert np.array_equal(out[:h * w * 3].cpu().numpy().reshape(h, w, 3), exp)
        assert not out[h * w * 3:].any()
    finally:
        ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, None)


def test_kept_symbols_with_long_and_short_subsequences(env, monkeypatch):
    """a stream whose subsequences hold far more than 64 symbols in places (flat areas: 1- and 2-bit codes) and few elsewhere (noise):
    the copy and the second decode of the long ones share the output"""
    ctx, torch, dev = env
    from cniic_amd import _lib
    monkeypatch.setenv("CNIIC_HD_KEEP", "1")
    rng = np.random.default_rng(11)
    img = np.zeros((256, 384, 3), np.uint8)
    img[:, :128] = 40                                              # flat
    img[:, 128:256] = rng.integers(0, 256, (256, 128, 3))          # noise
    img[:, 256:] = (np.arange(128)[None, :, None] // 8 * 8).astype(np.uint8)   # steps
    ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, 0)
    try:
        for expr in ("delta", "hufman"):
            rc, data, _ = ctx.encode(expr, img)
            assert rc == 0
            rc, back = ctx.decode(expr, data)
            assert rc == 0 and np.array_equal(back, img), expr
    finally:
        ctx.set_opt(_lib.OPT_GPU_DECODE_MIN, None)
