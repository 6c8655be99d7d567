#!/usr/bin/env python3
"""bench.py -- headline benchmark of the cniic hot path on MI355X.

Metric (BASELINE.json): Mpixels/sec encode, `cluster-colors` K=256.
Workload at N=1: configs[1] = one 4096x4096 synthetic photo-like RGB image, resident in HBM;
a "step" is one full Codec::encode (count_freqs dedup -> K-means to convergence -> remap ->
Huffman), the encoded stream landing in an HBM buffer.  At N>1 every rank holds its own 4096x4096
image (weak scaling) and the ranks cluster the UNION of their pixels into one shared palette:
dense histograms and the K partial centroid sums are all-reduced over RCCL each iteration.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `roofline` is the K-means assign kernel (dominant kernel), timed
live with HIP events on the stream it runs on; `cpu_baseline` is the CPU restatement of the
reference algorithm (oracle mode R) on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=4096, help="image side (default: configs[1], 4096)")
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--max-iters", type=int, default=0, help="0 = to convergence, like the reference")
    ap.add_argument("--cpu-sample", type=int, default=2560, help="side of the crop timed on the CPU (0 = skip)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import cniic_amd
    from cniic_amd import _lib, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knob: CNIIC_BENCH_FORCE_SHARDED=1 runs the multi-GPU code path (process group, collectives) with one rank
    sharded = world > 1 or os.environ.get("CNIIC_BENCH_FORCE_SHARDED") == "1"
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

    W = H = args.size
    K = args.k
    expr = "cluster-colors(%d)" % K
    seed = synth.SEED0 + 2 + rank  # config 2 of SURVEY 8(d); one image per rank
    # one non-default stream shared by torch (collectives) and the library (kernels): stream order
    # is the only synchronisation between them
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = cniic_amd.Context(local_rank, stream=stream.cuda_stream)

    img = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    ctx.synth_image(_lib.SYNTH_PHOTO, seed, W, H, out=img)
    out = torch.empty(W * H * 4 + (1 << 20), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    if sharded:
        from cniic_amd.dist import ShardedClusterColors
        enc = ShardedClusterColors(ctx, K, dist, dev, max_iters=args.max_iters,
                                   collectives="native" if world == 1 and os.environ.get("CNIIC_COLLECTIVES", "native") == "native" else None)

        def step():
            return enc.encode(img, W, H, out)
    else:
        def step():
            rc, ln, st = ctx.encode(expr, img, w=W, h=H, out=out, max_iters=args.max_iters)
            return ln, st

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nbytes, st = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    npx_total = W * H * world
    value = npx_total * args.steps / dt / 1e6
    ms_per_step = dt / args.steps * 1e3

    # ---- roofline of the dominant kernel: K-means assign over the distinct colours (dedup form,
    # 10 algorithmic bytes per colour per launch: 4 B key + 4 B weight + 1 B label read + 1 B written)
    roofline = None
    U = 0
    if rank == 0:
        keys, counts = ctx.hist_rgb24(img, npx=W * H)
        U = int(keys.size)
        del keys, counts
        # one more encode of the same image with a HIP start/stop event pair attached to every assign dispatch
        # (hipExtLaunchKernelGGL, on the stream the kernel runs on: the kernel's own begin and end, without the
        # ~4 us of dispatch an event pair AROUND a launch adds): average over ALL launches of a real encode
        rc, ln, stp = ctx.encode(expr, img, w=W, h=H, out=out, max_iters=args.max_iters, flags=_lib.KM_PROFILE)
        ms_sum, launches = ctx.kernel_time("kmeans_rgbw_assign")
        launch_ms = ms_sum / max(1, launches)
        algo_bytes = 10.0 * U
        achieved = algo_bytes / (launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE), see DESIGN.md 6
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("size") == W and tj.get("K") == K and tj.get("unique_colours") == U:
                traffic = tj.get("hbm_bytes_per_launch")
        roofline = {"kernel": "k_rgbw_assign_cells", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "launch_ms": round(launch_ms, 5), "launches": int(launches), "algorithmic_bytes_per_launch": algo_bytes,
                    "note": "HIP start/stop events on each of the %d assign dispatches of one encode (%d iterations + launches that exit "
                            "on the device-side done flag, as rocprofv3 --stats counts them); exact cell-pruned assign over %d distinct "
                            "colours, K=%d; algorithmic bytes = 10 B/colour (SURVEY 8(d) dedup form); traffic = PMC "
                            "2*FETCH_SIZE+WRITE_SIZE of a steady-state launch (profiles/traffic.json)" % (launches, stp["iterations"], U, K)}

    # ---- CPU baseline: oracle mode R (reference algorithm restated) on a bounded crop, 1 thread
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
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

    if rank == 0:
        line = {
            "metric": "Mpixels/sec encode (cluster-colors K=%d)" % K, "value": round(value, 3), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u64", "data": "synthetic",
            "config": {"workload": "cluster-colors(%d) encode of one %dx%d photo-like synthetic RGB image per GPU "
                                   "(seed 0x636E696963+2+rank), to convergence" % (K, W, H),
                       "pixels_per_gpu": W * H, "unique_colours": U, "kmeans_iterations": int(st["iterations"]),
                       "centroids_tested_per_colour_per_iteration": round(st["pair_evals"] / max(1, st["iterations"]) / max(1, U), 2),
                       "bytes_per_px": round(nbytes / (W * H), 4),
                       "parallelism": "1 GPU" if not sharded else "pixels sharded over %d GPUs (each keeps its own image's colours), shared palette: RCCL "
                                                                   "all-reduce of the colour occupancy (8 MiB, once) and of the K partial sums per iteration (%s)"
                                                                   % (world, "library communicator, in-stream" if enc.collectives == "native"
                                                                      else "torch.distributed")},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()  # rank 0 did the roofline encode on its own: tear down together
    if sharded:
        enc.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
