#!/usr/bin/env python3
"""bench.py -- headline benchmark of the cniic hot path on MI355X.

Metric (BASELINE.json): Mpixels/sec encode, `cluster-colors` K=256.

  --config c2 (default, EVERY N)  configs[1]: one 4096x4096 synthetic photo-like RGB image per GPU, resident in HBM; a
                                "step" is one full Codec::encode (count_freqs dedup -> K-means to convergence -> remap ->
                                Huffman), the stream landing in an HBM buffer.  At N > 1 the per-GPU workload is the same
                                (weak scaling: one 4096^2 image per GPU) and the N images share ONE palette: the colour
                                occupancy is all-reduced once and the K partial centroid sums every iteration (RCCL,
                                in-stream) -- so the driver's 1/2/4/8 curve compares like with like.
  --config c4                   configs[3]: a batch of 1920x1080 frames, --frames-per-gpu F of them per GPU (128: 1024 frames
                                on 8 GPUs), ONE palette for the whole batch: every rank partitions its own frames' pixels,
                                the colour occupancy is all-reduced once and the K partial centroid sums every iteration
                                (RCCL, in-stream); each frame is then its own Hufman stream.  Weak scaling: per-GPU work is
                                fixed, the palette is the union's.  A step = the whole batch encode.  EVERY default line also
                                carries this workload as a block of its own (`c4_one_gpu` at N = 1, `c4` at N > 1 with the
                                same run's one-rank timing of rank 0's frames and `efficiency_vs_one_gpu`).
  --config c3                   configs[2]: voronoi(2048) on one 4096x4096 image per GPU (its own metric; roofline = the 5-D
                                assign per iteration, 7 B/px).  Not the default anywhere.
  --config c5                   configs[4]: `delta` on one 16384x16384 image per GPU (its own metric: Mpixels/sec encode
                                (delta); roofline = the gather kernel, 3 B/px read).  Not the default anywhere.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `roofline` is the K-means assign kernel (dominant kernel), timed live with HIP events
attached to every dispatch on the stream it runs on; `cpu_baseline` is the CPU restatement of the reference algorithm
(oracle mode R) on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FRAME_W, FRAME_H = 1920, 1080


def traffic_for(key):
    """HBM bytes per launch of a block's dominant kernel from the PMC passes kept in profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE, separate runs, 2 x FETCH + WRITE: tools/make_traffic.sh), or None when that kernel has not been measured."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        e = t.get(key) if key else t
        return e.get("hbm_bytes_per_launch") if e else None
    except Exception:
        return None


def roofline_from_timers(ctx, U, K, note, traffic=None):
    """roofline of the assign kernel from the per-dispatch event timers of the LAST profiled run on ctx.
    Algorithmic bytes per launch = 10 B per distinct colour (dedup form: 4 B key + 4 B weight + 1 B label read
    + 1 B written).  `frac` is over the WORKING launches (one per iteration, and the one that finishes the last iteration);
    launches past convergence, which exit on the device-side flag, are listed apart."""
    ms_p, n_p = ctx.kernel_time("kmeans_rgbw_persist")
    if n_p:
        # the loop as ONE launch (k_kmeans_persist.hip): a launch processes U colours in every one of its iterations, so its
        # algorithmic bytes are 10 B x U x iterations; the duration is the dispatch's own (HIP events attached to it)
        _, n_it = ctx.kernel_time("kmeans_rgbw_persist_iters")
        iters = n_it / n_p
        algo = 10.0 * U * iters
        launch_ms = ms_p / n_p
        achieved = algo / (launch_ms * 1e-3) / 1e9
        return {"kernel": "k_rgbw_persist", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "launch_ms": round(launch_ms, 5), "launches": int(n_p),
                "iterations_per_launch": round(iters, 2), "us_per_iteration": round(launch_ms * 1e3 / max(1.0, iters), 3),
                "algorithmic_bytes_per_launch": algo, "algorithmic_bytes_per_iteration": 10.0 * U,
                "note": note + "; the whole K-means loop is ONE persistent launch whose points stay in LDS: algorithmic bytes = 10 B x U per iteration x "
                               "the launch's iterations, HBM traffic (`traffic`) is a small fraction of that by construction"}
    ms_all, n_all = ctx.kernel_time("kmeans_rgbw_assign")
    ms_w, n_w = ctx.kernel_time("kmeans_rgbw_assign_working")
    if not n_w:
        return None
    algo = 10.0 * U
    launch_ms = ms_w / n_w
    achieved = algo / (launch_ms * 1e-3) / 1e9
    by = {}
    for cls in ("first", "full", "skip", "final-update", "no-op"):
        ms, n = ctx.kernel_time("kmeans_rgbw_assign_" + cls)
        if n:
            by[cls] = {"launches": int(n), "us": round(ms * 1e3 / n, 2), "GBps_algorithmic": round(algo / (ms / n * 1e-3) / 1e9, 1)}
    return {"kernel": "k_rgbw_assign_cells", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "launch_ms": round(launch_ms, 5), "launches": int(n_w),
            "algorithmic_bytes_per_launch": algo, "no_op_launches": int(n_all - n_w),
            "all_launches": {"launches": int(n_all), "launch_ms": round(ms_all / max(1, n_all), 5)},
            "by_class": by, "note": note}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)    # (the driver's own choice; a step is 2 ms.  With 5 / 1 the first run on a fresh box read 2.28 ms:
    ap.add_argument("--warmup", type=int, default=5)    #  the timed region began while the card was still coming up -- same kernels, same launch times)
    ap.add_argument("--config", choices=["auto", "c2", "c3", "c4", "c5"], default="auto",
                    help="auto = c2 at every N (same per-GPU workload, so the 1/2/4/8 curve is one workload); c4: the frame batch; c3: voronoi(2048) 4096^2; c5: `delta` 16384^2")
    ap.add_argument("--c5-size", type=int, default=16384, help="c5: image side (default: configs[4], 16384)")
    ap.add_argument("--frames-per-gpu", type=int, default=128, help="c4: 1920x1080 frames per GPU (128 x 8 GPUs = the 1024 of configs[3])")
    ap.add_argument("--size", type=int, default=4096, help="c2: image side (default: configs[1], 4096)")
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--max-iters", type=int, default=0, help="0 = to convergence, like the reference")
    ap.add_argument("--cpu-sample", type=int, default=2560, help="side of the crop timed on the CPU (0 = skip every CPU leg)")
    ap.add_argument("--no-extras", action="store_true", help="c2 at N=1: skip the c4_one_gpu and host-buffer legs")
    ap.add_argument("--decode", action="store_true", help="c2 / c5: time Codec::decode of the stream the encode produced (stream and image resident in HBM) "
                                                           "instead of the encode; its own metric (Mpixels/sec decode)")
    args = ap.parse_args()

    # stdout carries ONE JSON line: everything else a library prints there (RCCL's version banner when a communicator is made)
    # goes to stderr -- file descriptor 1 points at stderr until the line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    import cniic_amd
    from cniic_amd import _lib, synth
    from cniic_amd.dist import ShardedClusterColors

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knob: CNIIC_BENCH_FORCE_SHARDED=1 runs the multi-GPU code path (process group, collectives) with one rank
    sharded = world > 1 or os.environ.get("CNIIC_BENCH_FORCE_SHARDED") == "1"
    config = args.config if args.config != "auto" else "c2"
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rehearsal knobs for a 1-GPU box: CNIIC_BENCH_BACKEND=gloo CNIIC_BENCH_ONE_GPU=1 put every rank on cuda:0
        if os.environ.get("CNIIC_BENCH_ONE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("CNIIC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    K = args.k
    expr = "cluster-colors(%d)" % K
    # one non-default stream shared by torch (collectives) and the library (kernels): stream order
    # is the only synchronisation between them
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = cniic_amd.Context(local_rank, stream=stream.cuda_stream)
    native = "native" if world == 1 and os.environ.get("CNIIC_COLLECTIVES", "native") == "native" else None

    def coll_desc(c):
        return {"native": "library communicator, in-stream", "mailbox": "one-shot exchange over IPC-mapped mailboxes, in-stream, instead of RCCL",
                "host": "the library's loop over a host transport"}.get(c, "torch.distributed")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, warmup, steps, reduce_max=True):
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = step()
        if reduce_max:
            barrier()
        else:
            torch.cuda.synchronize()   # a rank timing work of its own: its own clock, nobody else's
        dt = time.perf_counter() - t0
        if dist is not None and reduce_max:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, r

    def make_frames(F, first_frame):
        fr = torch.empty((F, FRAME_H, FRAME_W, 3), dtype=torch.uint8, device=dev)
        for f in range(F):   # frame f of the batch uses seed + f (config 4)
            ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 4 + first_frame + f, FRAME_W, FRAME_H, out=fr[f])
        return fr

    def run_c4(enc, F, warmup, steps, profile, reduce_max=True):
        """-> (seconds for `steps` batch encodes, bytes of this rank's streams, stats, distinct colours of this rank, roofline)"""
        frames = make_frames(F, rank * F)
        stride = FRAME_W * FRAME_H  # 1 byte per pixel between streams: K = 256 labels need at most 8 bits each + the tree
        out = torch.empty(stride * F, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        dt, (lens, st) = timed(lambda: enc.encode_frames(frames, FRAME_W, FRAME_H, F, out, stride), warmup, steps, reduce_max)
        # the timed streams against the oracle's (configs[3]'s one-GPU share: one palette over these frames, every frame's own stream)
        g4 = golden("c4")
        run_c4.parity = None
        if g4 and rank == 0 and world == 1 and enc.dist is None and (g4.get("frames"), g4.get("w"), g4.get("h")) == (F, FRAME_W, FRAME_H) and K == 256 and not args.max_iters:
            import hashlib
            hs = [hashlib.sha256(out[f * stride:f * stride + lens[f]].cpu().numpy().tobytes()).hexdigest() for f in range(F)]
            allh = hashlib.sha256("".join(hs).encode()).hexdigest()
            run_c4.parity = {"case": "c4", "all_frames_sha256": allh, "matches_oracle": bool(allh == g4.get("all_frames_sha256") and int(st["iterations"]) == g4.get("iterations"))}
        roof, U = None, 0
        if profile:
            keys, counts = ctx.hist_rgb24(frames, npx=F * FRAME_W * FRAME_H)
            U = int(keys.size)
            del keys, counts
            # one more batch encode with the per-dispatch timers on -- by THIS rank alone (the other ranks are not in it, so it must not
            # be a collective: a one-rank session over this rank's own frames; same kernels, this GPU's colours, its own palette)
            solo = ShardedClusterColors(ctx, K, None, dev, max_iters=args.max_iters)
            solo.flags = _lib.KM_PROFILE
            solo.encode_frames(frames, FRAME_W, FRAME_H, F, out, stride)
            solo.close()
            roof = roofline_from_timers(ctx, U, K, "HIP start/stop events on every assign dispatch of one batch encode of rank 0's frames alone; exact cell-pruned assign "
                                        "over this GPU's %d distinct colours (of %d frames), K=%d; algorithmic bytes = 10 B/colour/launch (SURVEY 8(d) dedup form)" % (U, F, K),
                                        traffic_for("c4"))
        return dt, int(sum(lens)), st, U, roof

    def golden(case):
        """the oracle's digest of this workload (tests/golden/fullsize_digests.json, made by tests/golden/make_fullsize_digests.py), or None"""
        try:
            with open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")) as f:
                return json.load(f)["cases"].get(case)
        except Exception:
            return None

    def digest_check(case, stream, nbytes, w, h, iterations=None):
        """-> {"oracle_digest": ..., "matches": bool} when rank 0's workload is one the oracle was run on at full size"""
        g = golden(case)
        if not g or rank != 0 or (g.get("w"), g.get("h")) != (w, h):
            return None
        import hashlib
        got = hashlib.sha256(stream[:nbytes].cpu().numpy().tobytes()).hexdigest()
        ok = got == g.get("sha256") and int(nbytes) == g.get("length") and (iterations is None or g.get("iterations") in (None, int(iterations)))
        return {"case": case, "stream_sha256": got, "matches_oracle": bool(ok)}

    def run_c3(size, warmup, steps):
        """configs[2]: voronoi(2048), 5-D position + colour K-means on one size^2 image per GPU (replicas), to convergence.
        roofline = k_xy_assign per iteration, SURVEY 8(d): 7 B/px/iteration (3 B pixel + u16 label read and written)."""
        W = H = size
        Kv = 2048
        img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 3 + rank, W, H, out=img)
        out = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

        def step():
            rc, ln, st = ctx.encode("voronoi(%d)" % Kv, img, w=W, h=H, out=out, max_iters=args.max_iters, allow=(_lib.FEW_ACTIVE,))
            return ln, st
        dt, (nbytes, st) = timed(step, warmup, steps)
        if rank != 0:
            return None
        rc, ln, stp = ctx.encode("voronoi(%d)" % Kv, img, w=W, h=H, out=out, max_iters=args.max_iters, flags=_lib.KM_PROFILE, allow=(_lib.FEW_ACTIVE,))
        ms, n = ctx.kernel_time("kmeans_xyrgb_iter")
        roofline = None
        if n:
            it_ms = ms / n
            algo = 7.0 * W * H
            traffic3 = None   # PMC passes over a whole run (tools/make_profiles.sh): mean HBM bytes per k_xy_assign launch, 2 * FETCH_SIZE + WRITE_SIZE
            try:
                with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                    t3 = json.load(f).get("c3")
                if t3 and t3.get("size") == W and t3.get("K") == Kv:
                    traffic3 = t3.get("hbm_bytes_per_launch")
            except Exception:
                pass
            roofline = {"kernel": "k_xy_assign (+ k_xy_update)", "bound": "hbm", "achieved": round(algo / (it_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round(algo / (it_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "traffic": traffic3, "launch_ms": round(it_ms, 5),
                        "launches": int(n), "algorithmic_bytes_per_launch": algo,
                        "note": "HIP events around the whole K-means loop of one more encode / its %d iterations (assign + update launches); "
                                "algorithmic bytes = 7 B/px/iteration (SURVEY 8(d)); tiles nothing changed for are skipped, so late iterations read less" % int(n)}
        cpu = None
        if args.cpu_sample > 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            s = min(768, W)
            crop = np.ascontiguousarray(img[:s, :s].cpu().numpy())
            t0 = time.perf_counter()
            rc, data, ost = O.encode("voronoi(%d)" % Kv, crop, mode=O.MODE_R)
            cdt = time.perf_counter() - t0
            cpu = {"value": round(s * s / cdt / 1e6, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                   "sample": "%dx%d crop of the same image, voronoi(%d), oracle mode R, %d iterations, %.1f s (rc %d)" % (s, s, Kv, ost.get("iterations", 0), cdt, rc)}
        return {"metric": "Mpixels/sec encode (voronoi K=%d)" % Kv, "value": round(W * H * world * steps / dt / 1e6, 3), "unit": "Mpixels/s",
                "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "i32/u64", "data": "synthetic",
                "config": {"workload": "configs[2]: voronoi(%d) encode (5-D position + colour K-means to convergence) of one %dx%d photo-like synthetic RGB "
                                       "image per GPU (seed 0x636E696963+3+rank); image and stream HBM-resident" % (Kv, W, H), "pixels_per_gpu": W * H,
                           "kmeans_iterations": int(st["iterations"]),
                           "centroids_tested_per_px_per_iteration": round(st["pair_evals"] / max(1, st["iterations"]) / (W * H), 2) if "pair_evals" in st else None,
                           "parallelism": "1 GPU" if world == 1 else "%d independent images, one per GPU (replicas, no collective)" % world},
                "roofline": roofline, "cpu_baseline": cpu, "parity": digest_check("v%d" % W, out, nbytes, W, H, st["iterations"])}

    def run_c5(size, warmup, steps):
        """configs[4]: `delta` (Hilbert gather + neighbour differences + symbol histogram + Huffman) on one size^2 image per GPU
        (independent images: replicas, no collective).  roofline = the gather kernel, SURVEY 8(d): 3 B/px read."""
        W = H = size
        img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        ctx.synth_image(_lib.SYNTH_PHOTO, synth.SEED0 + 5 + rank, W, H, out=img)
        out = torch.empty(W * H * 3 + (1 << 24), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

        def step():
            rc, ln, st = ctx.encode("delta", img, w=W, h=H, out=out)
            return ln, st
        dt, (nbytes, st) = timed(step, warmup, steps)
        if rank != 0:
            return None
        parity = digest_check("c5", out, nbytes, W, H)
        ctx.encode("delta", img, w=W, h=H, out=out, flags=_lib.KM_PROFILE)  # one more call with the stage timers (HIP events on the ctx stream)
        stages = {}
        for k in ("delta_gather", "delta_hist", "delta_tree", "huff_pack", "delta_finish"):   # in call order; together the whole call (delta_tree: compaction, sort, the host's merge with the GPU idle, codes)
            ms, n = ctx.kernel_time(k)
            if n:
                stages[k + "_ms"] = round(ms / n, 4)
        g_ms = stages.get("delta_gather_ms")
        roofline = None
        if g_ms:
            algo = 3.0 * W * H
            roofline = {"kernel": "k_delta_gather_p2", "bound": "hbm", "achieved": round(algo / (g_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(algo / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "traffic": traffic_for("c5") if (W, H) == (16384, 16384) else None, "launch_ms": g_ms, "launches": 1,
                        "algorithmic_bytes_per_launch": algo,
                        "with_symbol_stream": {"bytes_per_launch": 5.0 * W * H, "GBps": round(5.0 * W * H / (g_ms * 1e-3) / 1e9, 2),
                                               "frac": round(5.0 * W * H / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)},
                        "note": "HIP events around the gather launch of one more encode (stage timers); algorithmic bytes = 3 B/px read (SURVEY 8(d), "
                                "Hilbert gather + delta); the kernel also writes the 2 B/px symbol stream the later passes read: with_symbol_stream"}
        cpu = None
        if args.cpu_sample > 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            s = min(2048, W)
            crop = np.ascontiguousarray(img[:s, :s].cpu().numpy())
            t0 = time.perf_counter()
            rc, data, _ = O.encode("delta", crop)
            cdt = time.perf_counter() - t0
            cpu = {"value": round(s * s / cdt / 1e6, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                   "sample": "%dx%d crop of the same image, the CPU restatement of Delta::encode, %.1f s" % (s, s, cdt), "bytes_per_px": round(len(data) / (s * s), 4)}
        return {"metric": "Mpixels/sec encode (delta)", "value": round(W * H * world * steps / dt / 1e6, 3), "unit": "Mpixels/s", "n_gpus": world,
                "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u8/u32", "data": "synthetic",
                "config": {"workload": "configs[4]: delta encode (Hilbert gather + differences + histogram + Huffman) of one %dx%d photo-like synthetic RGB "
                                       "image per GPU (seed 0x636E696963+5+rank); image and stream HBM-resident" % (W, H), "pixels_per_gpu": W * H,
                           "bytes_per_px": round(nbytes / (W * H), 4),
                           "parallelism": "1 GPU" if world == 1 else "%d independent images, one per GPU (replicas, no collective)" % world},
                "roofline": roofline, "cpu_baseline": cpu, "stages": stages, "parity": parity}

    line = None
    if args.decode:
        line = bench_decode(args, ctx, torch, np, dev, rank, world, timed, config)
    elif config == "c4":
        F = args.frames_per_gpu
        enc = ShardedClusterColors(ctx, K, dist, dev, max_iters=args.max_iters, collectives=native)
        dt, nbytes, st, U, roof = run_c4(enc, F, args.warmup, args.steps, profile=(rank == 0))
        npx_total = F * FRAME_W * FRAME_H * world
        if rank == 0:
            cpu = None
            if world == 1 and args.cpu_sample > 0:
                cpu = cpu_all_cores(np, make_frames, expr)
            line = {
                "metric": "Mpixels/sec encode (cluster-colors K=%d)" % K, "value": round(npx_total * args.steps / dt / 1e6, 3), "unit": "Mpixels/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u64", "data": "synthetic",
                "config": {"workload": "configs[3]: cluster-colors(%d) over a batch of %d 1920x1080 photo-like synthetic frames (%d per GPU, frame f: seed "
                                       "0x636E696963+4+f), one palette for the whole batch, one Hufman stream per frame, to convergence; frames and streams HBM-resident" % (K, F * world, F),
                           "frames_per_gpu": F, "pixels_per_gpu": F * FRAME_W * FRAME_H, "unique_colours_rank0": U, "kmeans_iterations": int(st["iterations"]),
                           "bytes_per_px": round(nbytes / (F * FRAME_W * FRAME_H), 4),
                           "parallelism": "1 GPU" if world == 1 else "frames sharded over %d GPUs (each keeps its own frames' colours), shared palette: RCCL all-reduce of the "
                                          "colour occupancy (8 MiB, once) and of the K partial sums per iteration (%s)"
                                          % (world, coll_desc(enc.collectives))},
                "roofline": roof, "cpu_baseline": cpu, "parity": run_c4.parity,
            }
        enc.close()
    elif config == "c3":
        line = run_c3(args.size, args.warmup, args.steps)
    elif config == "c5":
        line = run_c5(args.c5_size, args.warmup, args.steps)
    else:
        W = H = args.size
        seed = synth.SEED0 + 2 + rank  # config 2 of SURVEY 8(d); one image per rank
        img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        ctx.synth_image(_lib.SYNTH_PHOTO, seed, W, H, out=img)
        out = torch.empty(W * H * 4 + (1 << 20), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        enc = None
        if sharded:
            enc = ShardedClusterColors(ctx, K, dist, dev, max_iters=args.max_iters, collectives=native)

            def step():
                return enc.encode(img, W, H, out)
        else:
            def step():
                rc, ln, st = ctx.encode(expr, img, w=W, h=H, out=out, max_iters=args.max_iters)
                return ln, st
        dt, (nbytes, st) = timed(step, args.warmup, args.steps)
        npx_total = W * H * world
        # the timed stream against the oracle's digest at this size (one image, its own palette: the un-sharded step only)
        parity = digest_check("c2", out, nbytes, W, H, st["iterations"]) if not sharded and K == 256 and not args.max_iters else None
        roofline, cpu, extras, U = None, None, {}, 0
        enc_collectives = enc.collectives if enc is not None else None
        if enc is not None:
            enc.close()
            enc = None
        if rank == 0:
            keys, counts = ctx.hist_rgb24(img, npx=W * H)
            U = int(keys.size)
            del keys, counts
            # one more encode of the same image with a HIP start/stop event pair attached to every assign dispatch
            # (hipExtLaunchKernelGGL, on the stream the kernel runs on: the kernel's own begin and end, without the
            # ~4 us of dispatch an event pair AROUND a launch adds)
            rc, ln, stp = ctx.encode(expr, img, w=W, h=H, out=out, max_iters=args.max_iters, flags=_lib.KM_PROFILE)
            # PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/make_traffic.sh): the persistent launch's bytes, or -- when the loop
            # ran as one launch per iteration -- a full-schedule launch's (the file's top level)
            persisted = ctx.kernel_time("kmeans_rgbw_persist")[1] > 0
            traffic = traffic_for("c2_persist") if persisted else traffic_for(None)
            roofline = roofline_from_timers(ctx, U, K, "HIP start/stop events on every assign dispatch of one encode (%d iterations); exact cell-pruned assign over %d "
                                            "distinct colours, K=%d; algorithmic bytes = 10 B/colour/iteration (SURVEY 8(d) dedup form); traffic = PMC 2*FETCH_SIZE+WRITE_SIZE "
                                            "of the launch (profiles/traffic.json)" % (stp["iterations"], U, K), traffic)
        def headline():
            ln = {
                "metric": "Mpixels/sec encode (cluster-colors K=%d)" % K, "value": round(npx_total * args.steps / dt / 1e6, 3), "unit": "Mpixels/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u64", "data": "synthetic",
                "config": {"workload": "configs[1]: cluster-colors(%d) encode of one %dx%d photo-like synthetic RGB image per GPU "
                                       "(seed 0x636E696963+2+rank), to convergence; image and stream HBM-resident" % (K, W, H),
                           "pixels_per_gpu": W * H, "unique_colours": U, "kmeans_iterations": int(st["iterations"]),
                           "centroids_tested_per_colour_per_iteration": round(st["pair_evals"] / max(1, st["iterations"]) / max(1, U), 2) if "pair_evals" in st else None,
                           "bytes_per_px": round(nbytes / (W * H), 4),
                           "parallelism": "1 GPU" if not sharded else "pixels sharded over %d GPUs (each keeps its own image's colours), shared palette: RCCL "
                                                                       "all-reduce of the colour occupancy (8 MiB, once) and of the K partial sums per iteration (%s)"
                                                                       % (world, coll_desc(enc_collectives))},
                "roofline": roofline, "cpu_baseline": cpu,
            }
            if parity is not None:
                ln["parity"] = parity
            return ln

        # The headline is measured.  Everything below is optional blocks -- at N > 1 code that has never run on more than one device
        # (ADVICE r03): a watchdog guarantees the line.  If the blocks are not done in time, rank 0 writes the headline with what was
        # finished and every rank leaves (rank 0 first: its peers may be stuck in a collective with it).
        import threading
        limit_s = float(os.environ.get("CNIIC_BENCH_EXTRAS_LIMIT_S", "300"))

        class Progress:
            """What the watchdog may read: JSON snapshots of the finished blocks and the name of the running one, under a lock
            (the timer thread used to walk dictionaries the main thread was still filling)."""
            def __init__(self, base):
                self.lock = threading.Lock()
                self.base = json.dumps(base)
                self.done = {}
                self.current = None

            def begin(self, name):
                with self.lock:
                    self.current = name

            def put(self, name, value):
                extras[name] = value
                snap = json.dumps(value)   # (a value that cannot be serialised fails HERE, on the main thread, inside the block's try)
                with self.lock:
                    self.done[name] = snap
                    self.current = None

            def rebase(self, base):
                snap = json.dumps(base)
                with self.lock:
                    self.base = snap

        prog = Progress(headline())

        def fire():
            # a block hung (a GPU kernel or a collective that never returns): the line with what was finished, the stuck block's
            # name, and a NON-ZERO exit -- a process stuck on the GPU must not read as a success
            try:
                if rank == 0:
                    with prog.lock:
                        ln = json.loads(prog.base)
                        for k_, v_ in prog.done.items():
                            ln[k_] = json.loads(v_)
                        cur = prog.current
                    ln["extras_error"] = "watchdog: the optional blocks did not finish within %.0f s (stuck in: %s); the line carries what had; exit code 3" % (limit_s, cur)
                    os.write(json_fd, (json.dumps(ln) + "\n").encode())
            finally:
                os._exit(3)
        watchdog = threading.Timer(limit_s + (0.0 if rank == 0 else 10.0), fire)
        watchdog.daemon = True
        watchdog.start()
        if rank == 0 and world == 1 and not sharded:
            if not args.no_extras:
                try:
                    # the trait the reference calls is host image -> host bytes (bench.rs:33-35): the same encode with both buffers in host memory
                    prog.begin("host_io_ms_per_step")
                    himg = img.cpu().numpy()
                    hout = np.empty(W * H * 2, np.uint8)   # (the caller's Vec<u8>, reused like a harness would)
                    ctx.encode(expr, himg, out=hout)
                    t0 = time.perf_counter()
                    for _ in range(5):
                        ctx.encode(expr, himg, out=hout)
                    prog.put("host_io_ms_per_step", round((time.perf_counter() - t0) / 5 * 1e3, 3))
                    del himg, hout
                    # configs[3] on this one GPU (the N > 1 lines carry the same block over N GPUs, with their own one-rank timing)
                    F = args.frames_per_gpu
                    prog.begin("c4_one_gpu")
                    e4 = ShardedClusterColors(ctx, K, None, dev, max_iters=args.max_iters)
                    d4, nb4, st4, U4, roof4 = run_c4(e4, F, 1, 2, profile=True)
                    e4.close()
                    c4blk = {"workload": "configs[3] on one GPU: %d frames 1920x1080, one palette, %d Hufman streams; frames and streams HBM-resident" % (F, F),
                                            "value": round(F * FRAME_W * FRAME_H * 2 / d4 / 1e6, 3), "unit": "Mpixels/s", "ms_per_step": round(d4 / 2 * 1e3, 3),
                                            "kmeans_iterations": int(st4["iterations"]), "unique_colours": U4, "bytes_per_px": round(nb4 / (F * FRAME_W * FRAME_H), 4),
                                            "roofline": roof4, "parity": run_c4.parity}
                    if args.cpu_sample > 0:
                        c4blk["cpu_baseline"] = cpu_all_cores(np, make_frames, expr)
                    prog.put("c4_one_gpu", c4blk)
                    prog.begin("batch_own_palettes")
                    # the reference's OWN batch semantics (bench.rs:24-35): the same frames, one palette EACH -- F independent encodes in
                    # one call (cniic_codec_encode_batch: the images dealt to worker streams); the all-cores CPU leg above is this workload
                    frames = make_frames(F, 0)
                    stride = FRAME_W * FRAME_H
                    outb = torch.empty(stride * F, dtype=torch.uint8, device=dev)
                    torch.cuda.synchronize()
                    best = None
                    for streams in (8, 16):
                        ctx.set_opt(_lib.OPT_BATCH_STREAMS, streams)
                        db, (rcb, lensb, rcsb, stsb) = timed(lambda: ctx.encode_batch(expr, frames, FRAME_W, FRAME_H, F, outb, stride, max_iters=args.max_iters), 1, 2)
                        v = F * FRAME_W * FRAME_H * 2 / db / 1e6
                        if best is None or v > best["value"]:
                            best = {"workload": "the reference's batch semantics: %d frames 1920x1080, one palette EACH (%d independent Codec::encode calls dealt to %d worker "
                                                "streams of one context), to convergence" % (F, F, streams), "value": round(v, 3), "unit": "Mpixels/s",
                                    "ms_per_step": round(db / 2 * 1e3, 3), "ms_per_frame": round(db / 2 / F * 1e3, 4), "worker_streams": streams,
                                    "kmeans_iterations_mean": round(sum(s_["iterations"] for s_ in stsb) / F, 1), "bytes_per_px": round(sum(lensb) / (F * stride), 4)}
                        best.setdefault("by_streams", {})[str(streams)] = round(v, 3)
                    ctx.set_opt(_lib.OPT_BATCH_STREAMS, None)
                    prog.put("batch_own_palettes", best)
                    del frames, outb
                    # configs[2] and configs[4] as blocks of the same line (every BASELINE config driver-timed); the
                    # same functions that make the --config c3 / c5 lines
                    for name, fn, size in (("c3", run_c3, 4096), ("c5", run_c5, 16384)):
                        try:
                            prog.begin(name)
                            blk = fn(size, 2, 6)   # (warm-up, steps: with 1 / 3 the c5 block read 1.45 .. 2.1 ms from run to run)
                            eb = {k: blk[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config", "roofline", "cpu_baseline", "parity") if k in blk}
                            if "stages" in blk:
                                eb["stages"] = blk["stages"]
                            prog.put(name, eb)
                        except Exception as e:
                            prog.put(name, {"error": "%s: %s" % (type(e).__name__, e)})
                    # the other half of the trait, driver-timed (bench.rs:45-46 decodes every image it encodes): the
                    # decode of configs[1]'s and configs[4]'s streams, each checked against its source (delta: equal; cluster-colors: the MSE)
                    for name, cfg in (("c2_decode", "c2"), ("c5_decode", "c5")):
                        try:
                            prog.begin(name)
                            a2 = argparse.Namespace(**vars(args))
                            a2.steps, a2.warmup = 6, 2
                            blk = bench_decode(a2, ctx, torch, np, dev, rank, world, timed, cfg)
                            eb = {k: blk[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "dtype", "config", "roofline", "cpu_baseline", "stages", "host_io_ms_per_step") if k in blk}
                            eb["round_trip"] = "decode(encode(img)) == img, checked on the device" if cfg == "c5" else "lossy codec: MSE against the source in config.mse_vs_source"
                            prog.put(name, eb)
                        except Exception as e:
                            prog.put(name, {"error": "%s: %s" % (type(e).__name__, e)})
                except Exception as e:   # (the headline above is measured: a failing extra is reported, not fatal)
                    prog.put("extras_error", "%s: %s" % (type(e).__name__, e))
            # ---- CPU baseline: oracle mode R (reference algorithm restated) on a bounded crop, 1 thread
            if args.cpu_sample > 0:
                prog.begin("cpu_baseline")
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_lib as O
                s = min(args.cpu_sample, W)
                crop = np.ascontiguousarray(img[:s, :s].cpu().numpy())
                t0 = time.perf_counter()
                rc, data, ost = O.encode(expr, crop, mode=O.MODE_R)
                cdt = time.perf_counter() - t0
                cpu = {"value": round(s * s / cdt / 1e6, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
                       "sample": "%dx%d crop of the same image, oracle mode R (reference algorithm incl. neighbour pruning), "
                                 "%d iterations, %.1f s" % (s, s, ost["iterations"], cdt),
                       "bytes_per_px": round(len(data) / (s * s), 4)}
                # the HIP path on the SAME crop (the two bytes/px figures were of different images)
                rcg, ng, stg = ctx.encode(expr, crop, max_iters=args.max_iters)
                cpu["hip_same_crop"] = {"bytes_per_px": round(len(ng) / (s * s), 4) if rcg == 0 else None, "iterations": int(stg["iterations"]) if rcg == 0 else None}
                prog.rebase(headline())
                prog.begin(None)
        if world > 1 and os.environ.get("CNIIC_BENCH_MAILBOX", "0") == "1" and enc_collectives != "mailbox":
            # OPT-IN (CNIIC_BENCH_MAILBOX=1): the exchange has only ever run between processes that share one GPU, and a fault
            # in a never-exercised cross-GPU path must not cost the driver its N > 1 line.
            # the same step with the K partial sums exchanged ONE-SHOT (every rank writes its sums into every peer's mailbox over the
            # direct xGMI links, k_mailbox.hip) instead of RCCL's ring; `value` above stays the RCCL figure.  Waits are bounded inside
            # the kernel (5 s here), a rank that cannot map its peers' mailboxes makes every rank skip the block.
            try:
                prog.begin("mailbox")
                os.environ["CNIIC_COLLECTIVE_TIMEOUT_MS"] = "5000"
                em = ShardedClusterColors(ctx, K, dist, dev, max_iters=args.max_iters, collectives="mailbox")
                if em.collectives == "mailbox":
                    dtm, (nbm, stm) = timed(lambda: em.encode(img, W, H, out), args.warmup, args.steps)
                    if rank == 0:
                        mbblk = {"what": "the same step, the per-iteration all-reduce of the K partial sums as a one-shot exchange over IPC-mapped mailboxes",
                                             "value": round(npx_total * args.steps / dtm / 1e6, 3), "unit": "Mpixels/s", "ms_per_step": round(dtm / args.steps * 1e3, 3),
                                             "kmeans_iterations": int(stm["iterations"]), "same_stream_as_rccl": bool(nbm == nbytes),
                                             "speedup_vs_default_collectives": round(dt / dtm, 4)}
                        prog.put("mailbox", mbblk)
                elif rank == 0:
                    prog.put("mailbox", {"unavailable": "the mailboxes could not be set up on every rank (IPC mapping or the known-answer exchange failed)"})
                em.close()
            except Exception as e:
                prog.put("mailbox", {"error": "%s: %s" % (type(e).__name__, e)})
            finally:
                os.environ.pop("CNIIC_COLLECTIVE_TIMEOUT_MS", None)
        if world > 1 and not args.no_extras:
            try:
                # configs[3] over these N GPUs (128 frames per GPU, one palette for all N x 128), and -- in the same run -- every rank's
                # own frames clustered by that rank ALONE (no collective; rank 0's is reported): the line is self-contained, its
                # efficiency does not lean on another invocation's N = 1 figure.
                F = args.frames_per_gpu
                prog.begin("c4")
                del img, out
                e4 = ShardedClusterColors(ctx, K, dist, dev, max_iters=args.max_iters, collectives=native)  # (its own communicator: the first one is closed)
                d4, nb4, st4, U4, roof4 = run_c4(e4, F, 1, 2, profile=False)
                coll4 = e4.collectives
                e4.close()
                solo = ShardedClusterColors(ctx, K, None, dev, max_iters=args.max_iters)
                d1, nb1, st1, U1, roof1 = run_c4(solo, F, 1, 2, profile=(rank == 0), reduce_max=False)
                solo.close()
                if rank == 0:
                    vN = F * FRAME_W * FRAME_H * world * 2 / d4 / 1e6
                    v1 = F * FRAME_W * FRAME_H * 2 / d1 / 1e6
                    c4n = {"workload": "configs[3]: %d frames 1920x1080 (%d per GPU) over %d GPUs, one palette, one Hufman stream per frame" % (F * world, F, world),
                                    "value": round(vN, 3), "unit": "Mpixels/s", "ms_per_step": round(d4 / 2 * 1e3, 3), "kmeans_iterations": int(st4["iterations"]),
                                    "collectives": coll4,
                                    "one_gpu_same_run": {"what": "rank 0's %d frames clustered by rank 0 alone (its own palette, no collective), timed in this run" % F,
                                                         "value": round(v1, 3), "ms_per_step": round(d1 / 2 * 1e3, 3), "kmeans_iterations": int(st1["iterations"]),
                                                         "unique_colours": U1, "roofline": roof1},
                                    "efficiency_vs_one_gpu": round(vN / (world * v1), 4)}
                    prog.put("c4", c4n)
            except Exception as e:   # the headline line above is already measured: an extra that fails must not take it down
                prog.put("c4", {"error": "%s: %s" % (type(e).__name__, e)})
        watchdog.cancel()
        if rank == 0:
            line = headline()
            line.update(extras)
        if enc is not None:
            enc.close()
    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.barrier()  # rank 0 did the roofline encode on its own: tear down together
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def bench_decode(args, ctx, torch, np, dev, rank, world, timed, config):
    """The other half of the trait (bench.rs:45-46 decodes every image it encodes): Codec::decode of the stream Codec::encode made,
    stream and image both resident in HBM (replicas: one image per GPU, no collective).  roofline = the kernel that decodes the
    payload and writes the symbols (k_hd_write): payload bytes read + the bytes it writes, timed with HIP events around it."""
    from cniic_amd import _lib, synth
    if config == "c5":
        W = H = args.c5_size
        expr, seed, what = "delta", synth.SEED0 + 5 + rank, "configs[4]"
    else:
        W = H = args.size
        expr, seed, what = "cluster-colors(%d)" % args.k, synth.SEED0 + 2 + rank, "configs[1]"
    lossless = expr == "delta"
    img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    ctx.synth_image(_lib.SYNTH_PHOTO, seed, W, H, out=img)
    stream = torch.empty(W * H * 3 + (1 << 24), dtype=torch.uint8, device=dev)
    rc, ln, st = ctx.encode(expr, img, w=W, h=H, out=stream, max_iters=args.max_iters)
    back = torch.empty(W * H * 3, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        return ctx.decode_into(expr, stream, ln, back)
    dt, (rc, dw, dh) = timed(step, args.warmup, args.steps)
    assert (dw, dh) == (W, H)
    if lossless:
        assert torch.equal(back.view(H, W, 3), img), "decode(encode(img)) != img"
    if rank != 0:
        return None
    mse = float((img.view(-1).to(torch.float32) - back.to(torch.float32)).pow(2).mean().item())   # bench::compute_error (bench.rs:95-104)
    ctx.set_opt(_lib.OPT_STAGE_TIMERS, 1)
    ctx.decode_into(expr, stream, ln, back)
    stages = {}
    for k in ("hd_pass0", "hd_check", "hd_write", "undiff_scatter"):
        ms, n = ctx.kernel_time(k)
        if n:
            stages[k + "_ms"] = round(ms / n, 4)
    ctx.set_opt(_lib.OPT_STAGE_TIMERS, None)
    himg = np.empty(W * H * 3, np.uint8)
    hstream = stream[:ln].cpu().numpy()
    ctx.decode_into(expr, hstream, ln, himg)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.decode_into(expr, hstream, ln, himg)
    host_ms = (time.perf_counter() - t0) / 3 * 1e3
    roofline = None
    w_ms = stages.get("hd_write_ms")
    if w_ms:
        sym_bytes = 4 if lossless else 3       # delta: packed differences (u32) for the prefix sum; the RGB codecs: pixels
        algo = float(ln) + float(sym_bytes) * W * H
        roofline = {"kernel": "k_hd_compact (+ the offsets scan in front of it; k_hd_write for the subsequences of more than 64 symbols)" if lossless
                              else "k_hd_write (+ the offsets scan in front of it)", "bound": "hbm", "achieved": round(algo / (w_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(algo / (w_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "traffic": traffic_for("c5_decode" if lossless else "c2_decode") if W == (16384 if lossless else 4096) else None, "launch_ms": w_ms, "launches": 1,
                    "algorithmic_bytes_per_launch": algo,
                    "note": "HIP events around the scan + write launches of one more decode (stage timers); algorithmic bytes = the %d-byte stream read once + "
                            "%d B/symbol written; the boundary passes before it (hd_pass0, hd_check) read the stream again and write 20 B per 1024 bits%s"
                            % (ln, sym_bytes, " -- and, for this long-coded stream, keep the symbols they meet: the write is a copy of those (k_hd_compact reads 4 B/symbol "
                                              "of kept rows instead of the stream; `traffic` is that kernel's)" if lossless else "")}
    cpu = None
    if args.cpu_sample > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        s = min(2048, W)
        crop = np.ascontiguousarray(img[:s, :s].cpu().numpy())
        rc, data, _ = ctx.encode(expr, crop, max_iters=args.max_iters)
        t0 = time.perf_counter()
        rco, dec = O.decode(expr, data)
        cdt = time.perf_counter() - t0
        cpu = {"value": round(s * s / cdt / 1e6, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
               "sample": "the CPU restatement's decode of the %dx%d crop's stream (%d bytes), %.1f s (rc %d)" % (s, s, len(data), cdt, rco)}
    return {"metric": "Mpixels/sec decode (%s)" % ("delta" if lossless else "cluster-colors K=%d" % args.k),
            "value": round(W * H * world * args.steps / dt / 1e6, 3), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u64", "data": "synthetic",
            "config": {"workload": "%s: %s DECODE of the stream its encode made of one %dx%d photo-like synthetic RGB image per GPU, stream (%d bytes) and image "
                                   "resident in HBM" % (what, expr, W, H, ln), "pixels_per_gpu": W * H, "stream_bytes": int(ln), "mse_vs_source": round(mse, 3),
                       "parallelism": "1 GPU" if world == 1 else "%d independent images, one per GPU (replicas, no collective)" % world},
            "roofline": roofline, "cpu_baseline": cpu, "stages": stages, "host_io_ms_per_step": round(host_ms, 3)}


def cpu_all_cores(np, make_frames, expr):
    """the reference's own batch semantics -- one image per rayon worker, one palette per image
    (bench.rs:24-35) -- as oracle mode R on every host core, one 1920x1080 frame per thread (ctypes releases the GIL)."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    O.lib()
    cores = max(1, min(os.cpu_count() or 1, 32))
    frames = make_frames(cores, 0).cpu().numpy()
    res = [None] * cores

    def work(i):
        res[i] = O.encode(expr, frames[i], mode=O.MODE_R)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    cdt = time.perf_counter() - t0
    npx = cores * FRAME_W * FRAME_H
    return {"value": round(npx / cdt / 1e6, 4), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "%d frames 1920x1080, one per thread, each with its own palette as the reference's harness does (oracle mode R), %.1f s" % (cores, cdt),
            "bytes_per_px": round(sum(len(r[1]) for r in res) / npx, 4)}


if __name__ == "__main__":
    main()
