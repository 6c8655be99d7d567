"""BASELINE.json's full sizes, HIP stream against the ORACLE's stream -- by digest.

tests/golden/fullsize_digests.json holds what oracle/ (mode L; oracle/kmeans_fast.c for the K-means, held to the plain loop by
tests/test_oracle_fast.py) produced for each workload in the build container: SHA-256, length, iteration count
(tests/golden/make_fullsize_digests.py).  Here the same workloads go through the C ABI on the GPU and must give the same bytes:
bit-exact parity at the sizes bench.py reports on, where tests/test_gpu_fullsize.py checks properties.
A case missing from the JSON (not generated yet) is skipped, never silently passed."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 0x636E696963


def golden(case):
    with open(os.path.join(HERE, "golden", "fullsize_digests.json")) as f:
        g = json.load(f)["cases"].get(case)
    if g is None:
        pytest.skip("tests/golden/fullsize_digests.json has no case %r yet" % case)
    return g


def sha(t, n):
    return hashlib.sha256(t[:n].cpu().numpy().tobytes()).hexdigest()


@pytest.fixture(scope="module")
def env():
    import torch

    import cniic_amd
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    yield ctx, torch, dev
    ctx.close()


def photo(ctx, torch, dev, seed, w, h):
    img = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    ctx.synth_image(1, seed, w, h, out=img)
    return img


@pytest.mark.parametrize("loop", ["persistent", "launches"])
def test_configs1_cluster_colors_256_at_4096(env, monkeypatch, loop):
    """configs[1]: the stream bench.py times, byte for byte the oracle's (61 iterations of exact Lloyd over 6.8 M colours), as ONE
    persistent launch (the default; heavy colours take the packed word's weight escape here -- the image has colours of more than
    254 pixels) and as one launch per iteration (CNIIC_OPT_KM_LOOP = 1)"""
    ctx, torch, dev = env
    from cniic_amd import _lib
    if loop == "persistent":
        monkeypatch.setenv("CNIIC_KM_PS_REQUIRE", "1")   # (testing build: a hand-over to the launches is an error)
    ctx.set_opt(_lib.OPT_KM_LOOP, 1 if loop == "launches" else None)
    try:
        _configs1(ctx, torch, dev)
    finally:
        ctx.set_opt(_lib.OPT_KM_LOOP, None)


def _configs1(ctx, torch, dev):
    g = golden("c2")
    img = photo(ctx, torch, dev, SEED + g["seed_offset"], g["w"], g["h"])
    assert hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest() == g["image_sha256"], "GPU generator != numpy generator"
    out = torch.empty(g["w"] * g["h"] * 2, dtype=torch.uint8, device=dev)
    rc, n, st = ctx.encode(g["codec"], img, w=g["w"], h=g["h"], out=out)
    assert rc == 0 and n == g["length"] and st["iterations"] == g["iterations"]
    assert sha(out, n) == g["sha256"]


def test_configs1_mode_r_band_at_4096(env):
    """K7 at full size: the reference's own pruned / truncated search (oracle mode R) on the same 4096^2 image -- the HIP path's
    bytes/px within 1 % and MSE within 2 % of it (tests/test_mode_r_band.py holds the band at 512^2 and 1536^2)"""
    ctx, torch, dev = env
    g = golden("c2r")
    img = photo(ctx, torch, dev, SEED + g["seed_offset"], g["w"], g["h"])
    out = torch.empty(g["w"] * g["h"] * 2, dtype=torch.uint8, device=dev)
    rc, n, st = ctx.encode(g["codec"], img, w=g["w"], h=g["h"], out=out)
    assert rc == 0
    back = torch.empty(g["w"] * g["h"] * 3, dtype=torch.uint8, device=dev)
    rc, dw, dh = ctx.decode_into(g["codec"], out, n, back)
    assert rc == 0
    mse = ctx.mse(img.cpu().numpy(), back.cpu().numpy())
    bpp = n / (g["w"] * g["h"])
    assert abs(bpp / g["bytes_per_px"] - 1) <= 0.01, (bpp, g["bytes_per_px"])
    assert abs(mse / g["mse"] - 1) <= 0.02, (mse, g["mse"])


def test_configs4_delta_at_16384(env):
    """configs[4]: `delta` on the 16384^2 image, 481 MB of stream, byte for byte the oracle's (scan: the build's frozen curve)"""
    ctx, torch, dev = env
    g = golden("c5")
    img = photo(ctx, torch, dev, SEED + g["seed_offset"], g["w"], g["h"])
    out = torch.empty(g["w"] * g["h"] * 3 + (1 << 24), dtype=torch.uint8, device=dev)
    rc, n, st = ctx.encode("delta", img, w=g["w"], h=g["h"], out=out)
    assert rc == 0 and n == g["length"]
    h = hashlib.sha256()
    for at in range(0, n, 1 << 26):
        h.update(out[at:min(n, at + (1 << 26))].cpu().numpy().tobytes())
    assert h.hexdigest() == g["sha256"]


def test_configs3_one_gpu_share_128_frames(env):
    """configs[3] as one GPU of the 8-GPU job sees it: 128 frames 1920 x 1080, ONE palette.  Every frame's stream (its own tree + its
    payload) and the stacked frames' single stream equal the oracle's: union clustering by exact Lloyd, then clusterc.rs:31-52 per frame."""
    ctx, torch, dev = env
    from cniic_amd.dist import ShardedClusterColors
    g = golden("c4")
    F, w, h, K = g["frames"], g["w"], g["h"], 256
    frames = torch.empty((F, h, w, 3), dtype=torch.uint8, device=dev)
    for f in range(F):
        ctx.synth_image(1, SEED + 4 + f, w, h, out=frames[f])
    stride = w * h
    out = torch.empty(stride * F, dtype=torch.uint8, device=dev)
    enc = ShardedClusterColors(ctx, K, None, dev)
    lens, st = enc.encode_frames(frames, w, h, F, out, stride)
    enc.close()
    assert st["iterations"] == g["iterations"]
    for f in range(F):
        want = g["frame_streams"][f]
        assert lens[f] == want["length"], "frame %d" % f
        assert sha(out[f * stride:], lens[f]) == want["sha256"], "frame %d" % f
    big = torch.empty(F * w * h + (1 << 20), dtype=torch.uint8, device=dev)
    rc, nb, stb = ctx.encode(g["codec"], frames, w=w, h=F * h, out=big)
    assert rc == 0 and nb == g["stacked_length"] and stb["iterations"] == g["iterations"]
    assert sha(big, nb) == g["stacked_sha256"]


@pytest.mark.parametrize("size", [1024, 2048, 4096])
def test_configs2_voronoi_2048(env, size):
    """configs[2]: voronoi(2048) -- 16 + 19 K bytes of centroids after exact Lloyd over every pixel in 5-D -- equal to the oracle's at
    1024^2, 2048^2 and the BASELINE size 4096^2 (6 x 10^12 distance evaluations on the CPU)"""
    ctx, torch, dev = env
    g = golden("v%d" % size)
    img = photo(ctx, torch, dev, SEED + g["seed_offset"], size, size)
    out = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    rc, n, st = ctx.encode(g["codec"], img, w=size, h=size, out=out)
    assert rc == 0 and n == g["length"] and st["iterations"] == g["iterations"]
    assert sha(out, n) == g["sha256"]


def test_the_release_library_gives_the_same_digests():
    """Every test above runs libcniic_hip_testing.so (tests/conftest.py); what a host links is libcniic_hip.so -- the same sources
    less -DCNIIC_TESTING.  One pass of this file against the RELEASE library, in a child process (the library is chosen at import):
    VERDICT r04, "the full-size digest tests should run once against the release .so too"."""
    import subprocess
    import sys
    if os.environ.get("CNIIC_USE_TESTING_LIB") == "0":
        pytest.skip("this IS the pass against the release library")
    env_ = dict(os.environ)
    env_["CNIIC_USE_TESTING_LIB"] = "0"
    for k in [k for k in env_ if k.startswith(("CNIIC_KM_", "CNIIC_TEST_", "CNIIC_DBG_"))]:
        del env_[k]   # (the release library would not read them anyway)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "not release_library", "-p", "no:cacheprovider"],
                       env=env_, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
    # and the child really had the release library: cniic_is_testing_build() == 0 there
    chk = subprocess.run([sys.executable, "-c", "import cniic_amd; print(cniic_amd.lib().cniic_is_testing_build())"], env=env_, cwd=os.path.dirname(HERE),
                         capture_output=True, text=True, timeout=300)
    assert chk.returncode == 0 and chk.stdout.strip().endswith("0"), chk.stdout + chk.stderr
