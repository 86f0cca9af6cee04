#!/usr/bin/env python3
"""Headline benchmark: fwd+bwd renders/s at 1 M Gaussians, 1920x1080, 8-view minibatch.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one view minibatch: every view is rendered forward and
backward (all five parameter gradients + the screen_coords gradient), the parameter gradients of
the rank's views are accumulated in one flat bucket, and with N > 1 ranks the gradients are exchanged
over RCCL (all-reduce of the 11 coords/scale/rotation/opacity floats per Gaussian, all-gather of the 3
colour-gradient floats per view from which every rank rebuilds the sh gradient).  The 8 views of the minibatch are split contiguously over the
ranks (strong scaling: the minibatch is fixed at 8 views).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# The view minibatch is pipelined over three HIP streams and the counter read-back uses a fourth: HIP's default of 4
# hardware queues per process (GPU_MAX_HW_QUEUES).  Streams that share a queue execute one after the other, so main()
# binds the pipeline's streams to the queues before the process group exists (gsplat_amd.minibatch.reserve_streams)
# and RCCL's streams share with them.  More queues do not help: with 8 or 16 the one-rank RCCL rehearsal measured
# 718 renders/s against 742 with 4, the plain run 775 with any.

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import gsplat_amd  # noqa: E402,F401
from gsplat_amd.camera import Camera  # noqa: E402
from gsplat_amd.minibatch import GradBucket, gather_colors, reduce_gradients, render_views, reserve_streams, shard_views  # noqa: E402
from gsplat_amd.loss import loss_and_grad  # noqa: E402
from gsplat_amd.rasterizer import StageTimer, ViewSpec, sh_grad_views  # noqa: E402
from gsplat_amd.scenes import synthetic_scene  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X vector fp32 spec


def minibatch_cameras(scene, n_views, device):
    """Cameras of one view minibatch: a ring of radius 0.15 around the scene's base pose, all
    looking down -z.  Every view sees (almost) the whole cloud, so the per-view work is uniform."""
    cams = []
    for v in range(n_views):
        ang = 2.0 * np.pi * v / n_views
        pos = np.array([0.15 * np.cos(ang), 0.15 * np.sin(ang), 0.0])
        cams.append(Camera(pos, np.array([1.0, 0.0, 0.0, 0.0]), scene.fx, scene.fy, scene.width / 2,
                           scene.height / 2, scene.width, scene.height, device=device))
    return cams


def algorithmic_bytes(n, m, pairs, pixels, k_sh, views=1, parts=1):
    """Compulsory HBM bytes of each stage: n Gaussians (m visible) with 236-byte parameter rows and 64-byte
    per-Gaussian outputs, `pairs` (tile, Gaussian) pairs at 44 bytes per compositing read, `pixels` pixels,
    k_sh active SH coefficients per channel.  All per view, except "preprocess_bwd_views": ONE launch chains the
    per-Gaussian sums of `views` views to the raw parameters - it reads the parameters and writes the dense
    gradients once per step, and reads 48 bytes of sums + the mask and writes the 12-byte screen gradient per view.
    "pair_reduce" reads the per-pair gradient records (48 bytes, `parts` rows per pair) and writes the sums.
    "preprocess_fwd_views" is stage 1 of all `views` views in ONE launch (the parameters are read once), here
    divided by the number of views so that it compares with a per-view time."""
    return {
        "preprocess_fwd": n * (12 + 12 + 16 + 4 + 12 * k_sh) + n + m * 64,
        "preprocess_fwd_views": (n * (12 + 12 + 16 + 4 + 12 * k_sh) + views * (n + m * 64)) / max(views, 1),
        "binning_sort": m * 16 + pairs * 12 + 2 * pairs * 12 + pairs * 8,
        "composite_fwd": pairs * (4 + 40) + pixels * 12 + pixels * 4,
        "composite_bwd": pairs * 44 + pixels * (12 + 12 + 8) + m * 40,
        "pair_reduce": pairs * parts * 48 + n * 9 + n * 48,
        "preprocess_bwd_views": n * (44 + 12 * k_sh) + n * 236 + views * n * (48 + 1 + 12),
    }


def executed_flops_bwd(stream_rec, num_pairs):
    """Flops the compositing backward EXECUTES for one view at tile 16, from the hint flags of its record stream:
    a half block (128 pixels) whose alpha is pinned at 0.01 takes the short path (transmittance, suffix and colour
    sums: about 20 flop per pixel), any other half the full one (about 60 flop per pixel and one exp2 + one rcp)."""
    flags = (stream_rec[:num_pairs].view(torch.int32)[:, 9] >> 30) & 3          # (one-block tiles: the hint bits ride above the slot)
    low_halves = int(((flags & 1) + ((flags >> 1) & 1)).sum().item())
    full_halves = 2 * num_pairs - low_halves
    return 128.0 * (60.0 * full_halves + 20.0 * low_halves), low_halves / max(2 * num_pairs, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--tile", type=int, default=16)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-loss", action="store_true",
                    help="dL/dimg of each view from the training loss 0.8*L1 + 0.2*(1-SSIM) against a synthetic "
                         "0..255 target (one more kernel per view) instead of a fixed cotangent")
    ap.add_argument("--no-overlap", action="store_true",
                    help="everything on one stream (kernel durations then are those of kernels running alone)")
    ap.add_argument("--forward-only", action="store_true",
                    help="render only, no backward and no collective (the forward-only stress configuration, e.g. "
                         "--gaussians 6000000 --width 3840 --height 2160)")
    ap.add_argument("--train-iter", action="store_true",
                    help="a full training iteration per step: --with-loss plus the densification statistics of every "
                         "view and one fused Adam step on the gradient bucket (Trainer.py:53-80 for a view minibatch)")
    args = ap.parse_args()
    if args.train_iter:
        args.with_loss = True

    if args.gpus > 1 and "RANK" not in os.environ:
        # started without the launcher: run it as a child (this process has not touched the GPU) and
        # leave with its exit code
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(29500 + os.getpid() % 2000), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    # Standard output carries the ONE JSON line and nothing else: whatever native libraries print while the run lasts
    # (RCCL announces its version on stdout when its communicator is created) goes to standard error instead.
    sys.stdout.flush()
    json_out = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GSR_BENCH_FORCE_DIST=1 rehearses the multi-rank code path (process group, all-reduce, barrier,
    # max over ranks) with a single rank, e.g. under torchrun --nproc-per-node 1
    use_dist = world > 1 or (os.environ.get("GSR_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the render path has no CPU implementation)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    reserve_streams(dev)       # before RCCL takes hardware queues
    if use_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.views % world != 0:
        raise SystemExit(f"--views {args.views} must be a multiple of the number of ranks {world}")

    n = args.gaussians
    scene = synthetic_scene(n, args.width, args.height, args.sh_degree, seed=0)
    params = [t.to(dev) for t in scene.params()]
    cams = minibatch_cameras(scene, args.views, dev)
    my_views = shard_views(args.views, world, rank)
    all_views = [ViewSpec.from_camera(c, args.sh_degree, args.tile) for c in cams]
    views = [all_views[v] for v in my_views]
    gen = torch.Generator().manual_seed(1)
    # dL/dimg of loss = mean over the minibatch of sum(img * R_v)
    cots = [(torch.rand(args.width, args.height, 3, generator=gen) / args.views).to(dev) for _ in range(args.views)]
    cots = [cots[v] for v in my_views]
    targets = None
    if args.with_loss:
        targets = [torch.floor(torch.rand(args.width, args.height, 3, generator=gen) * 256).to(dev) for _ in range(args.views)]
        targets = [targets[v] for v in my_views]

    # one flat fp32 bucket for the five parameter gradients (59 floats per Gaussian)
    bucket = GradBucket(n, dev)
    grads = list(bucket.views)

    # With several ranks the sh gradient (48 of the 59 floats per Gaussian) is not exchanged: the ranks
    # all-reduce the other 11 floats, all-gather 3 colour-gradient floats per view and rebuild it.
    g_colors = torch.empty(len(views), n, 3, dtype=torch.float32, device=dev) if use_dist and not args.forward_only else None

    stats = {}
    model = opt = None
    if args.train_iter:
        from gsplat_amd.densify import accumulate_stats_views, stats_delta
        from gsplat_amd.minibatch import reduce_densify_stats
        from gsplat_amd.model import GaussianModel
        from gsplat_amd.optimizer import Adam
        model = GaussianModel(*params, sh_degree=args.sh_degree)
        params = [model._coords.data, model._scale.data, model._rotation.data, model._opacity.data, model._sh.data]
        opt = Adam(params, lr=1e-3, betas=(0.5, 0.999))      # Trainer.py:15-16,24

    def cotangent(i, st):
        if targets is not None:    # loss = mean over the minibatch of the per-view training loss
            return loss_and_grad(st.img, targets[i], 0.2, 1.0 / args.views)[1]
        return cots[i]

    # (start, end) events on the current stream around the gradient exchange of the breakdown steps: the
    # stream waits there for the collectives, so the pair spans them and the sh rebuild
    exchange_events = None

    def step(overlap=True):
        # The rank's views are software-pipelined over three streams (minibatch.render_views): stage 1,
        # sorts and lists of the next views run beside the compositing of the current one; ONE pass then
        # chains all views to the raw parameters and writes the gradient bucket.
        # (with several ranks the gather of the colour gradients starts as soon as they exist, beside that pass)
        early = []
        states, _, g_screens = render_views(params, views, None if args.forward_only else cotangent,
                                            grads=tuple(grads), overlap=overlap and not args.no_overlap,
                                            g_colors=g_colors,
                                            colors_ready=lambda: early.append(gather_colors(g_colors, args.views, force_collectives=True)))
        if states and "m" not in stats:
            stats.update(m=states[0].num_visible, pairs=states[0].num_pairs)
            if states[0].num_visible and args.tile <= 16 and not args.forward_only:
                stats["executed"] = executed_flops_bwd(states[0].stream_rec, states[0].num_pairs)
        if args.forward_only:
            return
        if g_colors is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if exchange_events is not None else None
            if ev:
                ev[0].record()
            reduce_gradients(bucket, g_colors, args.views,
                             lambda g_all: sh_grad_views(all_views, params[0], params[4], list(g_all), bucket.sh),
                             force_collectives=True, gathered=early[0] if early else None)
            if ev:
                ev[1].record()
                exchange_events.append(ev)
        if opt is not None:
            live = [st for st in states if st.num_visible]
            if use_dist:
                # the statistics of the rank's views, summed over the ranks: every rank ends with those of
                # all views of the step (8 bytes per Gaussian exchanged)
                if live:
                    dg, dd = stats_delta(n, g_screens, [st.mask_u8 for st in live])
                else:
                    dg = torch.zeros((n, 1), dtype=torch.float32, device=dev)
                    dd = torch.zeros((n, 1), dtype=torch.int32, device=dev)
                reduce_densify_stats(model, dg, dd)
            else:
                accumulate_stats_views(model, g_screens, [st.mask_u8 for st in live])
            opt.step(grads=bucket.views)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    # Timed region: only the dominant kernel's entry point is bracketed by HIP events (on the stream it
    # is launched on); timing every call costs about 2.5 % of the step.
    dom = "composite_fwd" if args.forward_only else "composite_bwd"
    # (tile edges 32, 48 and 64 composite per-block streams: entry points gsr_composite_blocks_*)
    dom_names = ("gsr_" + dom, "gsr_" + dom.replace("composite_", "composite_blocks_"))
    with StageTimer(only=dom_names) as timer:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        elapsed = time.perf_counter() - t0
    dominant = timer.summary()
    # Per-stage breakdown: two more steps with every entry point timed, outside the timed region and on
    # ONE stream, so that every duration is that of the stage running alone.
    with StageTimer() as timer:
        exchange_events = [] if use_dist else None
        for _ in range(2):
            step(overlap=False)
        sync()
        exchange = exchange_events
        exchange_events = None
    stage = timer.summary()
    exchange_ms = sum(a.elapsed_time(b) for a, b in exchange) / len(exchange) if exchange else None
    breakdown_steps = 2
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        renders = args.views * args.steps
        value = renders / elapsed
        pixels = args.width * args.height
        k_sh = (args.sh_degree + 1) ** 2
        import ctypes
        from gsplat_amd import _lib as gsr_lib
        parts = int(gsr_lib.load().gsr_pairgrad_parts(ctypes.c_int(args.tile)))
        ab = algorithmic_bytes(n, stats.get("m", n), stats.get("pairs", 0), pixels, k_sh, views=max(len(views), 1), parts=parts)
        # device time of each stage per rendered view of this rank (the chain to the raw parameters runs
        # once per step for all views of the rank; pair_reduce + preprocess_bwd_views = stage-1 backward)
        views_timed = max(len(views), 1) * breakdown_steps
        stage_ms = {k.replace("gsr_", ""): v[1] / views_timed for k, v in stage.items()}
        for d in ("fwd", "bwd"):     # one name for the compositing stage, whichever entry point ran
            if "composite_blocks_" + d in stage_ms:
                stage_ms["composite_" + d] = stage_ms.get("composite_" + d, 0.0) + stage_ms.pop("composite_blocks_" + d)
        stage_ms["preprocess_bwd"] = stage_ms.get("pair_reduce", 0.0) + stage_ms.get("preprocess_bwd_views", 0.0)
        calls = sum(dominant.get(k, (0, 0.0))[0] for k in dom_names)
        total_ms = sum(dominant.get(k, (0, 0.0))[1] for k in dom_names) if calls else float("nan")
        dom_ms = total_ms / max(calls, 1)      # average launch duration over the timed region
        dom_alone_ms = stage_ms.get(dom)
        achieved = ab[dom] / (dom_ms * 1e-3) / 1e9
        # HBM-side bytes of one launch from the PMC passes under profiles/ - reported only when they were taken
        # on this workload's own first view (same pair count), otherwise null
        traffic = traffic_pairs = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            tj = json.load(open(tfile)).get(f"{n}_{args.width}x{args.height}_t{args.tile}_deg{args.sh_degree}", {})
            if tj.get("pairs") == stats.get("pairs"):
                traffic = tj.get("kernels", {}).get(dom, {}).get("total")
                traffic_pairs = tj.get("pairs")
        pairs = stats.get("pairs", 0)
        flops_bwd = 60.0 * pairs * 256.0 if args.tile == 16 else None
        out = {
            "metric": "fwd+bwd renders/sec @1M Gaussians 1080p, 1/2/4/8 GPUs; % HBM roofline",
            "value": round(value, 3),
            "unit": "renders/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{n} synthetic Gaussians, SH degree {args.sh_degree}, {args.width}x{args.height}, "
                            f"tile {args.tile}, {args.views}-view minibatch {'forward only' if args.forward_only else 'fwd+bwd'} split over {world} GPU(s), "
                            + ("RCCL all-reduce of 11N gradient floats + all-gather of 3N colour-gradient floats per view, sh gradient rebuilt on every rank"
                               if use_dist and not args.forward_only else "no collective")
                            + (", dL/dimg from the training loss kernel" if args.with_loss else "")
                            + (", densification statistics" + (" (summed over the ranks)" if use_dist else "") + " and fused Adam step"
                               if args.train_iter else ""),
                "gaussians": n, "width": args.width, "height": args.height, "tile_length": args.tile,
                "sh_degree": args.sh_degree, "views_per_step": args.views,
                "visible": stats.get("m"), "pairs_per_view": pairs,
            },
            "roofline": {
                "kernel": dom + "_kernel", "bound": "hbm",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_measured_at_pairs": traffic_pairs,
                "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": round(dom_ms, 4),
                "avg_launch_ms_alone": None if dom_alone_ms is None else round(dom_alone_ms, 4),
                "note": "compositing is bound by vector instruction issue at these list lengths (SQ counters: profiles/r04d_pmc_composite_sq.txt), not by HBM; "
                        "traffic = HBM-side bytes of one launch of this kernel on this workload's first view from the PMC passes under profiles/ (traffic.json); avg_launch_ms is measured in "
                        "the timed region, where the next view's pair emission, tile sort and record stream run beside the kernel on a second "
                        "stream and share the CUs with it (avg_launch_ms_alone: the same launch with nothing beside it); "
                        "stage 1 and the depth sort of all views of the step are one set of launches (stages *_views, per view here)",
            },
            "stage_ms_per_view": {k: round(v, 4) for k, v in stage_ms.items()},
            "exchange_ms_per_step": None if exchange_ms is None else round(exchange_ms, 4),
            # algorithmic bytes of a stage / its device time alone / 8 TB/s.  preprocess_bwd_views is ONE launch per
            # step for all views of the rank: its bytes and its time are both per step
            "stage_hbm_frac": dict(
                [(k, round(ab[k] / (stage_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
                 for k in ("preprocess_fwd", "preprocess_fwd_views", "composite_fwd", "composite_bwd", "pair_reduce") if stage_ms.get(k, 0.0) > 0.0] +
                ([("preprocess_bwd_views", round(ab["preprocess_bwd_views"] / (stage_ms["preprocess_bwd_views"] * max(len(views), 1) * 1e-3)
                                                 / 1e9 / HBM_PEAK_GBS, 4))] if stage_ms.get("preprocess_bwd_views", 0.0) > 0.0 else [])),
        }
        if flops_bwd and not args.forward_only:
            # 60 flop for every pixel of every pair (what the reference's formulation costs) ...
            out["roofline"]["valu_frac_algorithmic"] = round(flops_bwd / (dom_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 4)
            if "executed" in stats:
                # ... and the flops the kernel executes: halves of a block on which alpha is pinned at 0.01 take a short path
                out["roofline"]["valu_frac_executed"] = round(stats["executed"][0] / (dom_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 4)
                out["roofline"]["low_half_fraction"] = round(stats["executed"][1], 4)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, cams, args)
        sys.stdout.flush()
        os.write(json_out, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(scene, cams, args, budget_s=12.0):
    """The CPU port (oracle/) on this host's cores: forward + backward of the minibatch's views, one after
    the other, until the views are done or about ``budget_s`` seconds are spent (at least one view)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    threads = min(oracle.max_threads(), 16)
    cot = np.full((args.width, args.height, 3), 1.0 / args.views, np.float32)
    p = [t.numpy() for t in scene.params()]
    done = 0
    t0 = time.perf_counter()
    for cam in cams:
        oracle.render(p[0], p[1], p[2], p[3], p[4], args.sh_degree, args.width, args.height, args.tile,
                      cam.w2m_host, cam.proj_host, np.asarray(cam.transform.position, np.float32), dimg=cot, threads=threads)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 4), "unit": "renders/s", "cores": threads, "kind": "port",
            "sample": f"{done} of the {args.views} views, forward + backward, same scene and tile size ({dt:.1f} s)"}


if __name__ == "__main__":
    main()
