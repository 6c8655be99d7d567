"""kmeans::cluster::<ColorCount> as ONE persistent launch (cniic_amd/csrc/k_kmeans_persist.hip; reference src/kmeans.rs:21-39,
330-416, src/codec/clusterc.rs:68-114): the route every one-GPU cluster-colors(K <= 256) encode takes.  It must give the oracle's
run bit for bit -- on any grid size, with its points in LDS or in memory, when its barrier gives up (the launch-per-iteration loop
takes over from the untouched inputs), under the iteration cap, and through the codec.  CNIIC_KM_PS_REQUIRE makes a silent fall-back
an error, so "equal to the oracle" here is a statement about the persistent kernel."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from cniic_amd import Context
    c = Context(0)
    yield c
    c.close()


def synth_img(h, w, seed):
    from cniic_amd import synth
    return synth.photo(w, h, synth.SEED0 + seed)


def keys_of(img):
    p = img.reshape(-1, 3).astype(np.uint32)
    return (p[:, 0] << 16) | (p[:, 1] << 8) | p[:, 2]


def pts_of_keys(keys):
    return np.stack([(keys >> 16) & 255, (keys >> 8) & 255, keys & 255], axis=1).astype(np.int32)


def run_both(ctx, img, K, **kw):
    keys, counts = O.count_freqs(keys_of(img))
    w = counts.astype(np.uint32)
    rc, got = ctx.kmeans_rgbw(keys, w, K, **kw)
    rco, exp = O.kmeans(O.PT_RGBW, O.MODE_L, pts_of_keys(keys), w, K, max_iters=kw.get("max_iters", 0))
    assert rc == rco == 0
    assert got["stats"]["iterations"] == exp["stats"]["iterations"]
    assert np.array_equal(got["centroids"].astype(np.int32), exp["centroids"])
    assert np.array_equal(got["labels"], exp["labels"])
    assert np.array_equal(got["members"], exp["members"])
    return got, exp


@pytest.mark.parametrize("blocks", ["1", "3", "8", "37", "256"])
@pytest.mark.parametrize("K,shape", [(2, (32, 32)), (16, (64, 64)), (256, (128, 128)), (200, (300, 260))])
def test_any_grid_gives_the_oracles_run(ctx, monkeypatch, K, shape, blocks):
    if int(blocks) < 8 and shape[0] * shape[1] > 64 * 64:
        pytest.skip("more than 1024 cells a block: refused by design (test_a_range_with_too_many_cells_...)")
    monkeypatch.setenv("CNIIC_KM_PS_REQUIRE", "1")
    monkeypatch.setenv("CNIIC_KM_PS_BLOCKS", blocks)
    run_both(ctx, synth_img(*shape, seed=11 + K), K)


@pytest.mark.parametrize("lds_bytes", ["60000", "80000", "130000"])
@pytest.mark.parametrize("no_skip", [False, True])
def test_points_that_do_not_fit_lds_live_in_memory(ctx, monkeypatch, lds_bytes, no_skip):
    """a block's LDS budget shrunk until most of its cells keep their packed words in memory (45000: a thousand or two stay)"""
    from cniic_amd import _lib
    monkeypatch.setenv("CNIIC_KM_PS_REQUIRE", "1")
    monkeypatch.setenv("CNIIC_KM_PS_BLOCKS", "16")
    monkeypatch.setenv("CNIIC_TEST_PS_LDS_BYTES", lds_bytes)
    run_both(ctx, synth_img(300, 260, seed=5), 64, flags=_lib.KM_NO_SKIP if no_skip else 0)


@pytest.mark.parametrize("abort_at", ["1", "2", "7"])
def test_a_barrier_that_gives_up_hands_over_to_the_launch_per_iteration_loop(ctx, monkeypatch, abort_at):
    """CNIIC_TEST_PS_ABORT_AT: every block leaves at that barrier as if its wait had run out; nothing the classic loop reads was
    written, and it produces the oracle's run"""
    monkeypatch.delenv("CNIIC_KM_PS_REQUIRE", raising=False)
    monkeypatch.setenv("CNIIC_TEST_PS_ABORT_AT", abort_at)
    run_both(ctx, synth_img(128, 128, seed=3), 32)
    monkeypatch.setenv("CNIIC_KM_PS_REQUIRE", "1")      # ... and with REQUIRE the same abort is an error: the knob does what it says
    from cniic_amd import _lib
    keys, counts = O.count_freqs(keys_of(synth_img(128, 128, seed=3)))
    rc, _ = ctx.kmeans_rgbw(keys, counts.astype(np.uint32), 32, allow=(_lib.HIP,))
    assert rc == _lib.HIP


def test_a_range_with_too_many_cells_is_refused_before_any_block_starts(ctx, monkeypatch):
    """one block for an image of several thousand non-empty cells: k_ps_ranges raises the fail word, the launch leaves at once"""
    from cniic_amd import synth
    img = synth.uniform(256, 256, synth.SEED0 + 9)           # uniform noise: ~all 32768 cells occupied
    monkeypatch.delenv("CNIIC_KM_PS_REQUIRE", raising=False)
    monkeypatch.setenv("CNIIC_KM_PS_BLOCKS", "1")
    run_both(ctx, img, 16, max_iters=6)
    monkeypatch.setenv("CNIIC_KM_PS_REQUIRE", "1")           # (and with REQUIRE the refusal is an error: the fail word did its work)
    from cniic_amd import _lib
    keys, counts = O.count_freqs(keys_of(img))
    rc, _ = ctx.kmeans_rgbw(keys, counts.astype(np.uint32), 16, max_iters=6, allow=(_lib.HIP,))
    assert rc == _lib.HIP


@pytest.mark.parametrize("max_iters", [1, 2, 5])
def test_iteration_cap(ctx, monkeypatch, max_iters):
    monkeypatch.setenv("CNIIC_KM_PS_REQUIRE", "1")
    got, exp = run_both(ctx, synth_img(160, 160, seed=21), 48, max_iters=max_iters)
    assert got["stats"]["iterations"] == max_iters


def test_loop_option_selects_the_launch_per_iteration_loop(ctx, monkeypatch):
    from cniic_amd import _lib
    img = synth_img(200, 200, seed=8)
    monkeypatch.setenv("CNIIC_TEST_PS_ABORT_AT", "1")       # would abort the persistent launch ...
    monkeypatch.setenv("CNIIC_KM_PS_REQUIRE", "1")          # ... and make that an error
    ctx.set_opt(_lib.OPT_KM_LOOP, 1)                        # ... but the option keeps the launch away
    try:
        run_both(ctx, img, 40)
    finally:
        ctx.set_opt(_lib.OPT_KM_LOOP, None)


@pytest.mark.parametrize("sp_min", ["0", str(1 << 40)])
@pytest.mark.parametrize("K", [5, 64, 256])
def test_codec_streams_equal_the_oracles(ctx, monkeypatch, K, sp_min):
    monkeypatch.setenv("CNIIC_KM_PS_REQUIRE", "1")
    monkeypatch.setenv("CNIIC_SP_MIN_PIXELS", sp_min)
    img = synth_img(240, 320, seed=K)
    rc, data, st = ctx.encode("cluster-colors(%d)" % K, img)
    rco, edata, est = O.encode("cluster-colors(%d)" % K, img, mode=O.MODE_L)
    assert rc == rco == 0 and data == edata and st["iterations"] == est["iterations"]


def test_heavy_pixel_counts_take_the_escape(ctx, monkeypatch):
    """colours with 255 and more pixels: the packed word's 8-bit count is an escape to the count array"""
    monkeypatch.setenv("CNIIC_KM_PS_REQUIRE", "1")
    rng = np.random.default_rng(5)
    keys = np.sort(rng.choice(1 << 24, 3000, replace=False)).astype(np.uint32)
    w = rng.integers(1, 100000, keys.size).astype(np.uint32)
    w[::7] = 255
    w[1::7] = 254
    rc, got = ctx.kmeans_rgbw(keys, w, 50)
    rco, exp = O.kmeans(O.PT_RGBW, O.MODE_L, pts_of_keys(keys), w, 50)
    assert rc == rco == 0 and got["stats"]["iterations"] == exp["stats"]["iterations"]
    assert np.array_equal(got["centroids"].astype(np.int32), exp["centroids"]) and np.array_equal(got["labels"], exp["labels"])


def test_two_states_of_one_context_back_to_back_and_a_second_run(ctx, monkeypatch):
    """the barrier words, sums and exit record are per state / per launch: nothing leaks from one run into the next"""
    monkeypatch.setenv("CNIIC_KM_PS_REQUIRE", "1")
    for seed in (1, 2, 3):
        run_both(ctx, synth_img(96, 160, seed=seed), 24)


@pytest.mark.parametrize("sp_min", ["0", str(1 << 40)])
def test_an_encode_whose_unwatched_launch_gave_up_runs_the_launches_and_enqueues_the_labels_again(ctx, monkeypatch, sp_min):
    """Codec::encode does not wait for the persistent launch: it goes on to the pixels' labels and reads the verdict where it fetches
    the result block (km_rgbw_result_end).  A launch that gave up (CNIIC_TEST_PS_ABORT_AT) is followed by the launch-per-iteration loop
    and a second round of label kernels: the oracle's stream; with CNIIC_KM_PS_REQUIRE the same abort is an error"""
    from cniic_amd import _lib
    monkeypatch.setenv("CNIIC_SP_MIN_PIXELS", sp_min)
    monkeypatch.delenv("CNIIC_KM_PS_REQUIRE", raising=False)
    monkeypatch.setenv("CNIIC_TEST_PS_ABORT_AT", "2")
    img = synth_img(200, 280, seed=17)
    rc, data, st = ctx.encode("cluster-colors(48)", img)
    rco, edata, est = O.encode("cluster-colors(48)", img, mode=O.MODE_L)
    assert rc == rco == 0 and data == edata and st["iterations"] == est["iterations"]
    monkeypatch.setenv("CNIIC_KM_PS_REQUIRE", "1")
    rc, _, _ = ctx.encode("cluster-colors(48)", img, allow=(_lib.HIP,))
    assert rc == _lib.HIP
    monkeypatch.delenv("CNIIC_TEST_PS_ABORT_AT")
    rc, data, st = ctx.encode("cluster-colors(48)", img)          # ... and the context is fine afterwards
    assert rc == 0 and data == edata
