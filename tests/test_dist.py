"""The N>1 path: shared-palette cluster-colors driven by cniic_amd.dist.ShardedClusterColors.

CPU (gloo, world_size 2): the driver's protocol (occupancy all-reduce, every rank clustering its own
colours from their position in the list of all colours, per-iteration all-reduce of the K partial sums as
SIGNED DELTAS, per-rank Huffman of the reduced image) is run with the oracle as the compute backend and
must equal a single-process clustering of the union bit for bit.
GPU: the same driver on the HIP backend (world_size 1, and 2 ranks sharing the GPU over gloo)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def keys_of(img):
    p = img.reshape(-1, 3).astype(np.uint32)
    return (p[:, 0] << 16) | (p[:, 1] << 8) | p[:, 2]


def reseed_img(rank):
    """the colours of a forced-empty-cluster fixture (tests/golden/reseed_golden.npz) dealt to two ranks: the union is the fixture"""
    G = np.load(os.path.join(HERE, "golden", "reseed_golden.npz"))
    keys, wt = G["rgbw38_keys"][rank::2], G["rgbw38_weight"][rank::2]
    k = np.repeat(keys, wt.astype(np.int64))
    np.random.default_rng(rank).shuffle(k)
    return np.stack([(k >> 16) & 255, (k >> 8) & 255, k & 255], 1).astype(np.uint8).reshape(1, -1, 3)


def make_img(rank, h=40, w=56):
    from cniic_amd import synth
    if os.environ.get("TEST_RESEED_IMAGES") == "1":
        return reseed_img(rank)
    if "FUZZ_SEED0" in os.environ:  # tests/fuzz_dist.py: other sizes and seeds, the same in the spawned workers
        return synth.photo(int(os.environ["FUZZ_W"]), int(os.environ["FUZZ_H"]), 12345 + int(os.environ["FUZZ_SEED0"]) + rank)
    return synth.photo(w, h, synth.SEED0 + 40 + rank)


class OracleBackend:
    """CPU stand-in for HipBackend (tests only): same interface, compute done by the oracle."""

    def __init__(self):
        import torch
        self.torch = torch

    def hist_dense(self, img, npx):
        cnt = np.bincount(keys_of(np.asarray(img)), minlength=1 << 24).astype(np.int32)
        return self.torch.from_numpy(cnt)

    def new_partials(self, K):
        return self.torch.zeros(5 * K + 2, dtype=self.torch.int64)

    def occupancy(self, table):
        return self.torch.from_numpy((table.numpy() != 0).astype(np.int32))   # 0/1 per colour; the sum over ranks is > 0 where any has it

    def cc_create_local(self, table, occ, K, partials, max_iters=0, seed=0):
        """this rank's colours, placed in the ascending list of all occupied colours (the reference's point list)"""
        import oracle_lib as O
        cnt = table.numpy()
        keys = np.nonzero(cnt)[0].astype(np.uint32)                 # local points
        gkeys = np.nonzero(occ.numpy())[0].astype(np.uint32)        # all occupied colours
        U, Ug = keys.size, gkeys.size
        grank = np.searchsorted(gkeys, keys)
        unpack = lambda k: np.stack([(k >> 16) & 255, (k >> 8) & 255, k & 255], 1).astype(np.int32)
        st = dict(K=K, keys=keys, w=cnt[keys].astype(np.uint32), U=U, lo=0, hi=U, pts=unpack(keys), gpts=unpack(gkeys), partials=partials,
                  labels=O.init_labels(Ug, K)[grank], running=np.zeros(5 * K, np.int64), prev=np.zeros(5 * K, np.int64), it=0,
                  seed=seed or O.DEFAULT_SEED)
        ppc = Ug // K
        st["cent"] = np.stack([st["gpts"][Ug - (c + 1) * ppc] if c < K - 1 else st["gpts"][0] for c in range(K)]).astype(np.int32)
        return st

    def assign(self, st):
        import oracle_lib as O
        lo, hi, K = st["lo"], st["hi"], st["K"]
        if st.get("done"):     # iterations issued after convergence are no-ops
            return
        r = O.kmeans_step(O.PT_RGBW, st["pts"][lo:hi], st["w"][lo:hi], K, st["cent"], st["labels"][lo:hi])
        st["labels"][lo:hi] = r["labels"]
        full = np.concatenate([r["sums"].reshape(-1), r["wsum"], r["members"]]).astype(np.int64)
        p = st["partials"].numpy()
        p[:5 * K] = full - st["prev"]            # signed deltas (full sums at iteration 0)
        p[5 * K] = r["changed"]
        p[5 * K + 1] = 0
        st["prev"] = full

    def poll(self, st):
        return st["it"], st.get("done", False)

    def update(self, st):
        import oracle_lib as O
        K = st["K"]
        p = st["partials"].numpy()
        if st.get("done"):
            p[:] = 0
            return
        st["running"] += p[:5 * K]
        run = st["running"].astype(np.uint64)
        st["cent"], _ = O.kmeans_finalize(O.PT_RGBW, st["gpts"], K, st["seed"], st["it"], run[:3 * K].reshape(K, 3), run[3 * K:4 * K],
                                          run[4 * K:5 * K])   # (an empty cluster is reseeded from the list of ALL colours)
        st["it"] += 1
        if int(p[5 * K]) == 0:
            st["done"] = True
        p[:] = 0

    def finish(self, st, img, w, h, local_table, out):
        import oracle_lib as O
        img = np.asarray(img)
        lut = dict(zip(st["keys"].tolist(), st["labels"].tolist()))
        cent = st["cent"].astype(np.uint8)
        reduced = np.array([cent[lut[k]] for k in keys_of(img).tolist()], np.uint8).reshape(img.shape)
        rc, data, _ = O.encode("hufman", reduced)      # Hufman.encode(&reduced_img) clusterc.rs:52
        assert rc == 0
        out[:len(data)] = self.torch.frombuffer(bytearray(data), dtype=self.torch.uint8)
        return len(data), dict(iterations=st["it"])

    def finish_frames(self, st, frames, w, h, F, out, stride):
        frames = np.asarray(frames).reshape(F, h, w, 3)
        lens = []
        for f in range(F):
            n, _ = self.finish(st, frames[f], w, h, None, out[f * stride:])
            lens.append(n)
        return lens, dict(iterations=st["it"])

    def destroy(self, st):
        pass


def expected_streams(imgs, K):
    """single-process: oracle K-means (mode L) over the union histogram, then each image coded alone"""
    import oracle_lib as O
    allk = np.concatenate([keys_of(i) for i in imgs])
    keys, counts = O.count_freqs(allk)
    pts = np.stack([(keys >> 16) & 255, (keys >> 8) & 255, keys & 255], 1).astype(np.int32)
    rc, r = O.kmeans(O.PT_RGBW, O.MODE_L, pts, counts.astype(np.uint32), K)
    assert rc == 0
    lut = dict(zip(keys.tolist(), r["labels"].tolist()))
    cent = r["centroids"].astype(np.uint8)
    outs = []
    for img in imgs:
        reduced = np.array([cent[lut[k]] for k in keys_of(img).tolist()], np.uint8).reshape(img.shape)
        rc, data, _ = O.encode("hufman", reduced)
        outs.append(data)
    return outs, r["stats"]["iterations"]


def make_frames(rank, F, h, w):
    """this rank's F frames, one contiguous [F][h][w][3] array (frame f of the batch uses seed + f)"""
    from cniic_amd import synth
    return np.stack([synth.photo(w, h, synth.SEED0 + 4 + rank * F + f) for f in range(F)])


def _frames_worker(rank, world, port, K, use_hip, q, F, h, w, env=None):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.update(env or {})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cniic_amd.dist import ShardedClusterColors
        frames = make_frames(rank, F, h, w)
        stride = (w * h * 16 + 4096 + 3) & ~3
        if use_hip:
            import cniic_amd
            dev = torch.device("cuda", 0)
            torch.cuda.set_device(0)
            torch.cuda.set_stream(torch.cuda.Stream(device=dev))
            ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
            enc = ShardedClusterColors(ctx, K, dist, dev, collectives=(env or {}).get("TEST_COLLECTIVES"))
            if env and env.get("TEST_COLLECTIVES"):
                assert enc.collectives == env["TEST_COLLECTIVES"]
            tf = torch.from_numpy(frames).to(dev)
            out = torch.zeros(stride * F, dtype=torch.uint8, device=dev)
        else:
            enc = ShardedClusterColors(None, K, dist, None, backend=OracleBackend())
            tf = frames
            out = torch.zeros(stride * F, dtype=torch.uint8)
        lens, st = enc.encode_frames(tf, w, h, F, out, stride)
        host = out.cpu().numpy()
        q.put((rank, [bytes(host[f * stride:f * stride + lens[f]].tobytes()) for f in range(F)], int(st["iterations"])))
    finally:
        dist.destroy_process_group()


def _run_frames(world, K, use_hip, F, h, w, env=None):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frames_worker, args=(r, world, port, K, use_hip, q, F, h, w, env), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            r, data, it = q.get(timeout=180)
            res[r] = (data, it)
        for p in procs:
            p.join(60)
    finally:
        for p in procs:       # a rank that died leaves its peers in a collective: end them instead of waiting for gloo's timeout
            if p.is_alive():
                p.terminate()
                p.join(10)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return res


def _check_frames(res, world, K, F, h, w):
    allframes = [fr for r in range(world) for fr in make_frames(r, F, h, w)]
    exp, iters = expected_streams(allframes, K)
    for r in range(world):
        assert res[r][1] == iters
        for f in range(F):
            assert res[r][0][f] == exp[r * F + f], "rank %d frame %d: stream differs from the union clustering" % (r, f)


def test_frame_batch_gloo_world2_three_frames_per_rank():
    """north_star config 4 in small: two ranks x three frames, ONE palette over all six (the oracle clusters the union of
    their pixels in one process), six Hufman streams; frame size 45 x 31: no power of two, a label run that is not 16-byte aligned"""
    K, F, h, w = 8, 3, 31, 45
    _check_frames(_run_frames(2, K, False, F, h, w), 2, K, F, h, w)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, K, use_hip, q, env=None):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.update(env or {})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cniic_amd.dist import ShardedClusterColors
        img = make_img(rank)
        h, w = img.shape[:2]
        if use_hip:
            import cniic_amd
            dev = torch.device("cuda", 0)
            torch.cuda.set_device(0)
            torch.cuda.set_stream(torch.cuda.Stream(device=dev))
            ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
            enc = ShardedClusterColors(ctx, K, dist, dev, collectives=(env or {}).get("TEST_COLLECTIVES"))
            if env and env.get("TEST_COLLECTIVES"):
                assert enc.collectives == env["TEST_COLLECTIVES"]
            timg = torch.from_numpy(img).to(dev)
            out = torch.zeros(w * h * 16 + 4096, dtype=torch.uint8, device=dev)
        else:
            enc = ShardedClusterColors(None, K, dist, None, backend=OracleBackend())
            timg = img
            out = torch.zeros(w * h * 16 + 4096, dtype=torch.uint8)
        n, st = enc.encode(timg, w, h, out)
        if os.environ.get("TEST_RESEED_IMAGES") == "1":
            assert int(st["empty_reseeds"]) == int(os.environ["TEST_EXPECT_RESEEDS"]), st
        q.put((rank, bytes(out[:n].cpu().numpy().tobytes()), int(st["iterations"])))
    finally:
        dist.destroy_process_group()


def _run(world, K, use_hip, env=None):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, K, use_hip, q, env), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            r, data, it = q.get(timeout=120)
            res[r] = (data, it)
        for p in procs:
            p.join(60)
    finally:
        for p in procs:       # a rank that died leaves its peers in a collective: end them instead of waiting for gloo's timeout
            if p.is_alive():
                p.terminate()
                p.join(10)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return res


def test_sharded_driver_gloo_world2_matches_single_process():
    K = 8
    res = _run(2, K, use_hip=False)
    exp, iters = expected_streams([make_img(0), make_img(1)], K)
    for r in (0, 1):
        assert res[r][0] == exp[r], "rank %d stream differs" % r
        assert res[r][1] == iters


@pytest.mark.gpu
def test_sharded_hip_world1_equals_plain_encode():
    import torch

    import cniic_amd
    from cniic_amd.dist import ShardedClusterColors
    img = make_img(3, 96, 128)
    h, w = img.shape[:2]
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    rc, exp, st = ctx.encode("cluster-colors(16)", img)
    out = torch.zeros(w * h * 16, dtype=torch.uint8, device=dev)
    n, st2 = ShardedClusterColors(ctx, 16, None, dev).encode(torch.from_numpy(img).to(dev), w, h, out)
    assert bytes(out[:n].cpu().numpy().tobytes()) == exp and st2["iterations"] == st["iterations"]
    ctx.close()


@pytest.mark.gpu
def test_native_rccl_world1_equals_plain_encode():
    """the library's own RCCL communicator (nranks = 1) drives the in-stream loop: same bytes as the plain encode"""
    import torch
    import cniic_amd
    from cniic_amd.dist import ShardedClusterColors
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    img = make_img(0, 72, 88)
    h, w = img.shape[:2]
    out = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
    enc = ShardedClusterColors(ctx, 16, None, dev, collectives="native")
    assert enc.collectives == "native", "librccl could not be bound"
    n, st2 = enc.encode(torch.from_numpy(img).to(dev), w, h, out)
    # the communicator's all-reduce itself: in-place sum over one rank is the identity
    t = torch.arange(1000, dtype=torch.int64, device=dev)
    enc.be.comm_all_reduce(enc.comm, t)
    torch.cuda.synchronize()
    assert bool((t == torch.arange(1000, dtype=torch.int64, device=dev)).all())
    enc.close()
    got = out[:n].cpu().numpy().tobytes()
    out2 = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
    rc, n2, _ = ctx.encode("cluster-colors(16)", torch.from_numpy(img).to(dev), w=w, h=h, out=out2)
    assert got == out2[:n2].cpu().numpy().tobytes()
    ctx.close()


@pytest.mark.gpu
def test_sharded_hip_world2_shared_gpu_matches_single_process():
    """two ranks (sharing the one GPU of the test box; collectives over gloo) = the oracle's union result"""
    K = 8
    res = _run(2, K, use_hip=True)
    exp, iters = expected_streams([make_img(0), make_img(1)], K)
    for r in (0, 1):
        assert res[r][0] == exp[r], "rank %d stream differs" % r
        assert res[r][1] == iters


@pytest.mark.gpu
def test_sharded_hip_world2_through_the_pixel_partition(monkeypatch):
    """the same two ranks with every image taking the large-image route (cniic_cc_image_*): still the union's result"""
    K = 8
    res = _run(2, K, use_hip=True, env={"CNIIC_SP_MIN_PIXELS": "0"})
    exp, iters = expected_streams([make_img(0), make_img(1)], K)
    for r in (0, 1):
        assert res[r][0] == exp[r], "rank %d stream differs" % r
        assert res[r][1] == iters


@pytest.mark.gpu
def test_image_session_world1_equals_plain_encode(monkeypatch):
    """cniic_cc_image_begin / _occupancy / _create on one rank = ClusterColors::encode, both through the partition"""
    import torch
    import cniic_amd
    from cniic_amd.dist import ShardedClusterColors
    monkeypatch.setenv("CNIIC_SP_MIN_PIXELS", "0")
    img = make_img(5, 130, 517)
    h, w = img.shape[:2]
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    rc, exp, st = ctx.encode("cluster-colors(16)", img)
    out = torch.zeros(w * h * 16, dtype=torch.uint8, device=dev)
    n, st2 = ShardedClusterColors(ctx, 16, None, dev).encode(torch.from_numpy(img).to(dev), w, h, out)
    assert bytes(out[:n].cpu().numpy().tobytes()) == exp and st2["iterations"] == st["iterations"]
    monkeypatch.setenv("CNIIC_SP_MIN_PIXELS", str(1 << 40))   # and both equal the dense-table route
    rc, exp2, st3 = ctx.encode("cluster-colors(16)", img)
    assert exp2 == exp and st3["iterations"] == st["iterations"]
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["dense", "partition"])
def test_native_loop_world2_over_a_host_transport(route):
    """cniic_cc_run -- the library's own K-means loop, all-reduce between the assign launches, convergence read at a fixed
    place of the sequence -- with TWO ranks (sharing the test box's GPU): the communicator is the host-transport one
    (cniic_comm_create_host, here gloo), so the C loop sees real sums of two ranks.  Result = the oracle's union result."""
    K = 8
    env = {"TEST_COLLECTIVES": "host", "CNIIC_SP_MIN_PIXELS": "0" if route == "partition" else str(1 << 40)}
    res = _run(2, K, use_hip=True, env=env)
    exp, iters = expected_streams([make_img(0), make_img(1)], K)
    for r in (0, 1):
        assert res[r][0] == exp[r], "rank %d stream differs" % r
        assert res[r][1] == iters


@pytest.mark.gpu
@pytest.mark.parametrize("world,route", [(2, "dense"), (2, "partition"), (4, "partition")])
def test_native_loop_over_mailboxes(world, route):
    """cniic_cc_run with the ONE-SHOT exchange (cniic_comm_create_mailbox): two / four processes that share the test box's GPU map
    each other's mailboxes through HIP IPC (the handles travel over gloo); every iteration's K partial sums are written into
    every peer's mailbox and added in rank order, and the 8 MiB of occupancy nibbles go through the same mailboxes in pieces.
    Result = the oracle's union result, as over RCCL or a host transport.  (The exchange folded into the K-means launches, an
    opt-in of round 4 that could not run on a shared GPU at full occupancy, is archived: profiles/r05_pruned_variants.patch.)"""
    K = 8 if world == 2 else 16
    env = {"TEST_COLLECTIVES": "mailbox", "CNIIC_COLLECTIVE_TIMEOUT_MS": "20000",
           "CNIIC_SP_MIN_PIXELS": "0" if route == "partition" else str(1 << 40)}
    res = _run(world, K, use_hip=True, env=env)
    exp, iters = expected_streams([make_img(r) for r in range(world)], K)
    for r in range(world):
        assert res[r][0] == exp[r], "rank %d stream differs" % r
        assert res[r][1] == iters


@pytest.mark.gpu
def test_native_loop_world4_over_a_host_transport():
    """four ranks (occupancy nibbles summed over four images, four-way partial sums) through the library's own loop"""
    K = 16
    res = _run(4, K, use_hip=True, env={"TEST_COLLECTIVES": "host", "CNIIC_SP_MIN_PIXELS": "0"})
    exp, iters = expected_streams([make_img(r) for r in range(4)], K)
    for r in range(4):
        assert res[r][0] == exp[r], "rank %d stream differs" % r
        assert res[r][1] == iters


# ------------------------------------------------------------------ config 4: a batch of frames, one palette
@pytest.mark.gpu
@pytest.mark.parametrize("route", ["dense", "partition"])
@pytest.mark.parametrize("F,h,w", [(4, 48, 64), (3, 29, 37)])
def test_frame_batch_hip_one_rank_equals_union_clustering(monkeypatch, route, F, h, w):
    """one rank, F frames: cniic_cc_finish_frames gives, per frame, the stream of the oracle's union clustering
    (label runs aligned to 16 bytes and not; dense-table and pixel-partition routes)"""
    import torch
    import cniic_amd
    from cniic_amd.dist import ShardedClusterColors
    monkeypatch.setenv("CNIIC_SP_MIN_PIXELS", "0" if route == "partition" else str(1 << 40))
    K = 16
    frames = make_frames(0, F, h, w)
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    stride = (w * h * 4 + 4096 + 3) & ~3
    out = torch.zeros(stride * F, dtype=torch.uint8, device=dev)
    lens, st = ShardedClusterColors(ctx, K, None, dev).encode_frames(torch.from_numpy(frames).to(dev), w, h, F, out, stride)
    exp, iters = expected_streams(list(frames), K)
    host = out.cpu().numpy()
    assert st["iterations"] == iters
    for f in range(F):
        assert bytes(host[f * stride:f * stride + lens[f]].tobytes()) == exp[f], "frame %d" % f
        rc, back = ctx.decode("cluster-colors(%d)" % K, exp[f])
        assert rc == 0 and back.shape == (h, w, 3)
    # a host output buffer takes the staging route
    import ctypes as C
    from cniic_amd import _lib
    ctx.close()


def _encode_frames_hip(frames, K, env_host_trees):
    import torch
    import cniic_amd
    from cniic_amd.dist import ShardedClusterColors
    F, h, w = frames.shape[:3]
    old = os.environ.pop("CNIIC_FRAME_TREES_HOST", None)
    if env_host_trees:
        os.environ["CNIIC_FRAME_TREES_HOST"] = "1"
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
        ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        stride = (w * h * 4 + 8192 + 3) & ~3
        out = torch.zeros(stride * F, dtype=torch.uint8, device=dev)
        lens, st = ShardedClusterColors(ctx, K, None, dev).encode_frames(torch.from_numpy(frames).to(dev), w, h, F, out, stride)
        host = out.cpu().numpy()
        ctx.close()
        return [bytes(host[f * stride:f * stride + lens[f]].tobytes()) for f in range(F)], st
    finally:
        os.environ.pop("CNIIC_FRAME_TREES_HOST", None)
        if old is not None:
            os.environ["CNIIC_FRAME_TREES_HOST"] = old


@pytest.mark.gpu
@pytest.mark.parametrize("K", [2, 5, 64, 256])
def test_frame_codes_on_the_gpu_equal_the_host_and_the_oracle(monkeypatch, K):
    """the per-frame Huffman codes, code tables and stream headers made by k_frame_trees (K <= 256) against the host path
    (CNIIC_FRAME_TREES_HOST=1) and the oracle's union clustering, on a batch with awkward frames: one of a single colour (a
    one-symbol code: zero-length, huf.rs:140-142), one of two colours, one that uses few of the clusters, noisy ones"""
    monkeypatch.setenv("CNIIC_SP_MIN_PIXELS", "0")
    from cniic_amd import synth
    h, w = 37, 53
    rng = np.random.default_rng(1234 + K)
    frames = [synth.photo(w, h, synth.SEED0 + 40 + f) for f in range(3)]
    frames.append(np.full((h, w, 3), 200, dtype=np.uint8))                                            # one colour
    two = np.zeros((h, w, 3), dtype=np.uint8); two[:, w // 3:] = (250, 10, 40); frames.append(two)    # two colours
    frames.append(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8))                               # noise: every cluster in use
    few = synth.photo(w, h, synth.SEED0 + 77) // 64 * 64; frames.append(few.astype(np.uint8))         # a coarse palette
    frames = np.stack(frames)
    gpu, st_g = _encode_frames_hip(frames, K, False)
    host, st_h = _encode_frames_hip(frames, K, True)
    assert st_g["iterations"] == st_h["iterations"]
    for f in range(len(frames)):
        assert gpu[f] == host[f], "frame %d: GPU and host code construction differ" % f
    exp, iters = expected_streams(list(frames), K)
    assert st_g["iterations"] == iters
    for f in range(len(frames)):
        assert gpu[f] == exp[f], "frame %d differs from the oracle's stream" % f


@pytest.mark.gpu
def test_frame_batch_hip_world2_over_a_host_transport():
    """two ranks x three frames through the library's own loop (host-transport communicator, both ranks on the test box's GPU)"""
    K, F, h, w = 8, 3, 31, 45
    _check_frames(_run_frames(2, K, True, F, h, w, env={"TEST_COLLECTIVES": "host", "CNIIC_SP_MIN_PIXELS": "0"}), 2, K, F, h, w)


@pytest.mark.gpu
def test_two_image_sessions_open_on_one_context(monkeypatch):
    """two cniic_cc_image_begin sessions opened on ONE context before either K-means state is created (and a plain encode in
    between): each keeps its own colour count (it used to be read from a slot of the context that the later calls overwrite)"""
    import torch
    import cniic_amd
    from cniic_amd.dist import HipBackend
    monkeypatch.setenv("CNIIC_SP_MIN_PIXELS", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    be = HipBackend(ctx, dev)
    K = 16
    imgs = [make_img(7, 40, 56), make_img(8, 96, 160)]          # different numbers of distinct colours
    exp = [ctx.encode("cluster-colors(%d)" % K, im)[1] for im in imgs]
    t = [torch.from_numpy(im).to(dev) for im in imgs]
    hs = [be.image_begin(t[i], imgs[i].shape[0] * imgs[i].shape[1]) for i in range(2)]
    ctx.encode("cluster-colors(4)", make_img(9, 64, 64))        # one more count through the context in between
    got = []
    for i in (0, 1):
        occ = be.image_occupancy(hs[i])
        part = be.new_partials(K)
        be.image_create(hs[i], occ, K, part)
        be.run(hs[i], None)
        out = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
        n, _ = be.finish(hs[i], t[i], imgs[i].shape[1], imgs[i].shape[0], None, out)
        got.append(out[:n].cpu().numpy().tobytes())
        be.destroy(hs[i])
    assert got == exp
    ctx.close()


@pytest.mark.gpu
def test_frame_batch_hip_world2_over_mailboxes():
    """two ranks x three frames, the loop's all-reduce the one-shot exchange"""
    K, F, h, w = 16, 3, 40, 56
    env = {"TEST_COLLECTIVES": "mailbox", "CNIIC_COLLECTIVE_TIMEOUT_MS": "20000", "CNIIC_SP_MIN_PIXELS": "0"}
    _check_frames(_run_frames(2, K, True, F, h, w, env=env), 2, K, F, h, w)


# ------------------------------------------------------------------ a rank that fails must not strand its peers
def _failing_worker(rank, world, port, q, collectives="host"):
    import datetime
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), CNIIC_SP_MIN_PIXELS="0")
    if rank == 1:
        os.environ["CNIIC_TEST_FAIL_AT_LAUNCH"] = "3"       # fault injection: rank 1 fails before enqueuing its fourth assign launch
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=20))
    import cniic_amd
    from cniic_amd import _lib
    from cniic_amd.dist import ShardedClusterColors
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    ctx = cniic_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    enc = ShardedClusterColors(ctx, 8, dist, dev, collectives=collectives)
    assert enc.collectives == collectives
    img = make_img(rank)
    h, w = img.shape[:2]
    out = torch.zeros(w * h * 16 + 4096, dtype=torch.uint8, device=dev)
    code, msg = 0, ""
    try:
        enc.encode(torch.from_numpy(img).to(dev), w, h, out)
    except _lib.CniicError as e:
        code, msg = e.code, str(e)
    finally:
        aborted = bool(getattr(enc.be, "host_aborted", False))
        try:
            dist.destroy_process_group()                     # (rank 1 leaves: rank 0's pending all-reduce fails instead of waiting)
        except Exception:
            pass
    q.put((rank, code, aborted, msg))


@pytest.mark.gpu
def test_a_failing_rank_aborts_its_communicator_and_its_peer_errors_out():
    """rank 1 fails inside cniic_cc_run (injected before launch 3): it aborts its communicator -- the host transport's callback
    is told -- and returns the error; rank 0, already waiting in the all-reduce of that iteration, comes back with
    CNIIC_ERR_RCCL when the transport gives up on the missing peer, instead of hanging"""
    import torch.multiprocessing as mp
    from cniic_amd import _lib
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(2):
            r, code, aborted, msg = q.get(timeout=150)
            res[r] = (code, aborted, msg)
    finally:
        for p in procs:
            p.join(20)
            if p.is_alive():
                p.terminate()
    assert res[1][0] == _lib.HIP and res[1][1], res          # the injected failure, and the abort notification reached the transport
    assert res[0][0] == _lib.RCCL, res                       # the peer: an error, not a hang


@pytest.mark.gpu
def test_a_failing_rank_ends_its_peers_wait_in_the_mailbox_kernel():
    """the same with the one-shot exchange: rank 1 fails before its fourth launch and writes the abort word into rank 0's
    mailbox; rank 0's kernel, waiting for rank 1's slice, leaves at once and rank 0 returns CNIIC_ERR_RCCL"""
    import torch.multiprocessing as mp
    from cniic_amd import _lib
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q, "mailbox"), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(2):
            r, code, aborted, msg = q.get(timeout=150)
            res[r] = (code, msg)
    finally:
        for p in procs:
            p.join(20)
            if p.is_alive():
                p.terminate()
    assert res[1][0] == _lib.HIP, res
    assert res[0][0] == _lib.RCCL and "peer aborted" in res[0][1], res


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["dense", "partition"])
def test_native_loop_world2_empty_cluster_reseed(route):
    """the empty-cluster branch (kmeans.rs:117-134) with two ranks: the re-seed picks from the index of ALL ranks' colours
    (gidx_select in the folded-in update); images = a forced-reseed fixture dealt to the ranks, result = the fixture's"""
    G = np.load(os.path.join(HERE, "golden", "reseed_golden.npz"))
    K, (iters, reseeds, _) = int(G["rgbw38_K"][0]), (int(v) for v in G["rgbw38_stats"])
    assert reseeds >= 2
    env = {"TEST_COLLECTIVES": "host", "TEST_RESEED_IMAGES": "1", "TEST_EXPECT_RESEEDS": str(reseeds),
           "CNIIC_SP_MIN_PIXELS": "0" if route == "partition" else str(1 << 40)}
    res = _run(2, K, use_hip=True, env=env)
    os.environ["TEST_RESEED_IMAGES"] = "1"
    try:
        exp, it = expected_streams([make_img(0), make_img(1)], K)
    finally:
        del os.environ["TEST_RESEED_IMAGES"]
    assert it == iters
    for r in (0, 1):
        assert res[r][0] == exp[r] and res[r][1] == iters


# ------------------------------------------------------------------ the driver's N > 1 invocation of bench.py, rehearsed
@pytest.mark.gpu
def test_bench_two_ranks_same_workload_as_one_rank_and_self_contained_c4_block():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` as the driver launches it (two ranks sharing the test
    box's one GPU, rendezvous over gloo): ONE JSON line on stdout; the top-level workload is configs[1] PER GPU -- the workload
    of the N = 1 line, so the driver's 1/2/4/8 curve is one workload -- and the line carries configs[3] as a block of its own
    with the same run's one-rank timing and an efficiency that needs no other invocation."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, CNIIC_BENCH_BACKEND="gloo", CNIIC_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", CNIIC_USE_TESTING_LIB="0",   # bench.py runs on the release library
               CNIIC_BENCH_MAILBOX="1")                                    # ... and the opt-in mailbox block is asked for
    common = ["--steps", "1", "--warmup", "1", "--size", "1024", "--frames-per-gpu", "2", "--cpu-sample", "0"]
    p2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                        env=env, capture_output=True, text=True, timeout=600)
    assert p2.returncode == 0, p2.stderr[-3000:]
    lines2 = [l for l in p2.stdout.splitlines() if l.strip()]
    assert len(lines2) == 1, p2.stdout
    two = json.loads(lines2[0])
    p1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--no-extras"] + common, env=env, capture_output=True,
                        text=True, timeout=600)
    assert p1.returncode == 0, p1.stderr[-3000:]
    one = json.loads([l for l in p1.stdout.splitlines() if l.strip()][0])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["metric"] == one["metric"] and two["unit"] == one["unit"] and two["scaling"] == one["scaling"] == "weak"
    assert two["config"]["workload"] == one["config"]["workload"]            # the same per-GPU workload at every N
    assert two["config"]["pixels_per_gpu"] == one["config"]["pixels_per_gpu"] == 1024 * 1024
    assert two["value"] > 0 and two["cpu_baseline"] is None
    mb = two["mailbox"]     # the same step once more with the one-shot exchange (here: two processes mapping each other's mailboxes on one GPU)
    assert mb["value"] > 0 and mb["same_stream_as_rccl"] and mb["kmeans_iterations"] == two["config"]["kmeans_iterations"], mb
    c4 = two["c4"]
    assert c4["value"] > 0 and c4["one_gpu_same_run"]["value"] > 0
    assert abs(c4["efficiency_vs_one_gpu"] - c4["value"] / (2 * c4["one_gpu_same_run"]["value"])) < 1e-3
    assert c4["one_gpu_same_run"]["roofline"]["frac"] > 0
