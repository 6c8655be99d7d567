"""The drop-in boundary under the conditions the reference's harness creates (src/bench.rs:15-83): `encode` / `decode`
called concurrently from worker threads, one image per task (bench.rs:24-28; here one cniic_ctx per thread, as
INTEGRATION.md prescribes), and the harness itself -- tools/cniic_bench, the C++ counterpart of bench.rs -- run as a
program on configs[0] (`Hufman` on one 512 x 512 image) with its CSV checked column by column (bench.rs:68-75)."""
import os
import subprocess
import threading

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_threads_two_contexts_concurrently():
    """rayon's worker threads (bench.rs:24-28): every thread owns a context and runs every codec on its own images while
    the other does the same; each stream equals the one the same call gives alone, decodes round-trip"""
    from cniic_amd import Context, synth
    exprs = ["cluster-colors(16)", "delta", "hufman", "voronoi(8)", "hilbert(rle)"]
    imgs = {t: [synth.photo(160 + 16 * t, 120 + 8 * i, synth.SEED0 + 100 + 10 * t + i) for i in range(3)] for t in range(2)}
    with Context(0) as c0:
        alone = {t: [[c0.encode(e, im)[1] for e in exprs] for im in imgs[t]] for t in range(2)}
    got, errors = {}, []
    start = threading.Barrier(2)

    def worker(t):
        try:
            with Context(0) as ctx:
                start.wait()
                res = []
                for rep in range(4):
                    res = []
                    for im in imgs[t]:
                        row = []
                        for e in exprs:
                            rc, data, _ = ctx.encode(e, im)
                            rc2, back = ctx.decode(e, data)
                            assert rc == 0 and rc2 == 0
                            if e in ("delta", "hufman", "hilbert(rle)"):
                                assert np.array_equal(back, im)
                            row.append(data)
                        res.append(row)
                got[t] = res
        except Exception as ex:  # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(ex)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not errors, errors
    assert got == alone


def test_harness_program_config1_hufman_512(tmp_path):
    """configs[0]: `Hufman` on one 512 x 512 image through the harness program.  CSV of output/Hufman.csv: the reference's four
    columns (bench.rs:68-75) -- name, compressed_size, compression_ratio = size / (w h 24) x 100 (bench.rs:41-43, 74), error = MSE
    (0: a lossless codec whose decode differs is an error, bench.rs:50-59); stdout carries the four extra columns of SURVEY 5"""
    from cniic_amd import synth
    exe = os.path.join(ROOT, "tools", "cniic_bench")
    assert os.path.exists(exe), "tools/cniic_bench is built by `make -C cniic_amd/csrc`"
    r = subprocess.run([exe, "--codec=hufman", "synth:P:512x512:1"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().split("\n")
    assert lines[0] == "name,compressed_size,compression_ratio,error,mpix_per_s,iters,hbm_gbps,roofline_frac"
    cols = lines[1].split(",")
    assert cols[0] == "synth:P:512x512:1" and len(cols) == 8 and float(cols[4]) > 0 and float(cols[6]) > 0 and 0 <= float(cols[7]) < 1
    csv = open(tmp_path / "output" / "Hufman.csv").read().strip().split("\n")       # codec.name() (hufc.rs:43)
    assert csv[0] == "name,compressed_size,compression_ratio,error"
    name, size, ratio, err = csv[1].split(",")
    img = synth.photo(512, 512, synth.SEED0 + 1)
    rc, data, _ = O.encode("hufman", img)
    assert rc == 0 and name == "synth:P:512x512:1" and int(size) == len(data)
    assert abs(float(ratio) - len(data) / (512 * 512 * 24) * 100.0) < 1e-9 and float(err) == 0.0


def test_harness_program_config0_reads_png_files(tmp_path):
    """configs[0] as BASELINE.json states it: `Hufman` on one 512 x 512 RGB **PNG** through the harness (bench.rs:30 `image::open`).
    The harness's own PNG reader (zlib inflate + the five scanline filters) against PIL-written files: 8-bit RGB (with and without
    the optimiser, so that several filter types occur), RGBA (alpha dropped like `to_rgb8`), palette and grey -- the CSV's size is the
    oracle's for the decoded pixels and the lossless round trip holds (error 0)."""
    from PIL import Image
    from cniic_amd import synth
    exe = os.path.join(ROOT, "tools", "cniic_bench")
    rgb = synth.photo(512, 512, synth.SEED0 + 1)
    small = synth.photo(97, 61, synth.SEED0 + 2)
    files = {}
    Image.fromarray(rgb, "RGB").save(tmp_path / "c0.png"); files["c0.png"] = rgb
    Image.fromarray(rgb, "RGB").save(tmp_path / "c0_opt.png", optimize=True, compress_level=9); files["c0_opt.png"] = rgb
    rgba = np.dstack([small, np.full(small.shape[:2], 77, np.uint8)])
    Image.fromarray(rgba, "RGBA").save(tmp_path / "rgba.png"); files["rgba.png"] = small
    pal = Image.fromarray(small, "RGB").quantize(64)
    pal.save(tmp_path / "pal.png"); files["pal.png"] = np.asarray(pal.convert("RGB"))
    grey = small[..., 1].copy()
    Image.fromarray(grey, "L").save(tmp_path / "grey.png"); files["grey.png"] = np.dstack([grey] * 3)
    r = subprocess.run([exe, "--codec=hufman"] + sorted(files), cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rows = {l.split(",")[0]: l.split(",")[1:] for l in open(tmp_path / "output" / "Hufman.csv").read().strip().split("\n")[1:]}
    assert set(rows) == set(files)
    for name, img in files.items():
        rc, data, _ = O.encode("hufman", np.ascontiguousarray(img))
        assert rc == 0 and int(rows[name][0]) == len(data), name
        assert float(rows[name][2]) == 0.0
    (tmp_path / "bad.png").write_bytes(open(tmp_path / "c0.png", "rb").read()[:2000])      # a cut file: refused, not crashed on
    r = subprocess.run([exe, "--codec=hufman", "bad.png"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot read image" in r.stderr


def test_harness_program_two_images_on_worker_threads(tmp_path):
    """two images, two worker threads with a context each, a lossy codec: two CSV rows, sizes as the oracle's"""
    from cniic_amd import synth
    exe = os.path.join(ROOT, "tools", "cniic_bench")
    specs = ["synth:P:96x64:1", "synth:U:40x33:1"]
    r = subprocess.run([exe, "--codec=cluster-colors(16)"] + specs, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rows = dict(l.split(",")[0:1] + [l.split(",")[1:]] for l in open(tmp_path / "output" / "cluster-colors_16.csv").read().strip().split("\n")[1:])
    assert set(rows) == set(specs)
    for spec, img in ((specs[0], synth.photo(96, 64, synth.SEED0 + 1)), (specs[1], synth.uniform(40, 33, synth.SEED0 + 1))):
        rc, data, _ = O.encode("cluster-colors(16)", img, mode=O.MODE_L)
        rc2, back = O.decode("cluster-colors(16)", data)
        assert int(rows[spec][0]) == len(data)
        assert abs(float(rows[spec][2]) - O.mse(img, back)) <= 1e-6 * max(1.0, O.mse(img, back))
    r = subprocess.run([exe, "--codec=nonsense(3)", specs[0]], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "Malformed codec argument" in r.stderr                # codec.rs:41-59


@pytest.mark.gpu
def test_encode_batch_equals_separate_encodes():
    """cniic_codec_encode_batch = the harness's many-images loop (bench.rs:24-35) in one call: every image on its own -- its own
    palette, its own stream -- byte for byte what cniic_codec_encode gives for it, whatever the number of worker streams; a frame
    that fails (fewer colours than clusters) fails alone."""
    import torch
    import cniic_amd
    from cniic_amd import _lib, synth
    dev = torch.device("cuda", 0)
    F, w, h = 11, 160, 96
    frames = np.stack([synth.photo(w, h, synth.SEED0 + 200 + f) for f in range(F)])
    frames[7] = 33                                          # one colour: cluster-colors(16) cannot cluster it
    fr_d = torch.from_numpy(frames).to(dev)
    with cniic_amd.Context(0) as ctx:
        for expr in ("cluster-colors(16)", "delta", "hufman", "voronoi(8)"):
            single = [ctx.encode(expr, frames[f], allow=(_lib.TOO_FEW_POINTS, _lib.FEW_ACTIVE)) for f in range(F)]
            stride = w * h * 16 + 4096
            for streams in (1, 3, 8):
                ctx.set_opt(_lib.OPT_BATCH_STREAMS, streams)
                out = torch.zeros(stride * F, dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()   # (the context runs on a stream of its own: torch's fill must have landed)
                rc, lens, rcs, sts = ctx.encode_batch(expr, fr_d, w, h, F, out, stride, allow=(_lib.TOO_FEW_POINTS, _lib.FEW_ACTIVE))
                host = out.cpu().numpy()
                for f in range(F):
                    assert rcs[f] == single[f][0], (expr, f)
                    if rcs[f] == 0:
                        assert host[f * stride:f * stride + lens[f]].tobytes() == single[f][1], (expr, streams, f)
                        assert sts[f]["iterations"] == single[f][2]["iterations"]
                assert rc == next((r for r in rcs if r != 0), 0)
            ctx.set_opt(_lib.OPT_BATCH_STREAMS, None)
        # host buffers in and out
        hout = np.zeros(stride * F, np.uint8)
        rc, lens, rcs, _ = ctx.encode_batch("delta", frames, w, h, F, hout, stride)
        assert rc == 0 and all(hout[f * stride:f * stride + lens[f]].tobytes() == ctx.encode("delta", frames[f])[1] for f in range(F))
