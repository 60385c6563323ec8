#!/usr/bin/env python
"""Benchmark of the hot path (BASELINE.json metric): videos/sec of the Swin3D-T(GRPB) trunk + VQAHead
on synthetic 8-fragment x 32 x 224 x 224 clips ("video" = 8 clips), inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W           (N>1: launched by torch.distributed.run)

A "step" = one forward of the hot path over one batch of ``--batch`` clips per GPU (default 4 =
BASELINE.json configs[1], "C2").  Prints ONE JSON line on rank 0 with the contract's keys plus
``roofline`` (dominant kernel: algorithmic flops / hipEvent-measured launch time vs the dense
MFMA peak) and ``cpu_baseline`` (the CPU oracle timed on this box's host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense bf16/fp16, MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0
CLIPS_PER_VIDEO = 8
SWIN_T_GFLOP_PER_CLIP = 175.53   # SURVEY.md §8d (2*MAC, GEMM-only, padding as the reference pads)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the first ~20 steps (35 ms) after start-up run ~5 % slower (clocks / queues still ramping); 10 + 60 steps of
    # 1.7 ms measure the steady state a 900-video job sees and still finish in a blink
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="clips per GPU per step (C2: 4)")
    ap.add_argument("--dtype", default=os.environ.get("KVQ_OPERAND_DTYPE", "fp16"), choices=["fp16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=6)
    ap.add_argument("--overlap", choices=["batch", "steps"], default="steps",
                    help="with --streams > 1: 'batch' splits each step's clips over the streams, 'steps' sends whole "
                         "consecutive steps to alternating streams")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the steps are issued on (independent batches: the VALU-bound attention of one "
                         "can overlap the MFMA-bound GEMMs of another, launch gaps and tails are filled)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("KVQ_BENCH_GRAPH", "0")),
                    help="1: capture one step per stream in a hipGraph (static inputs, resident in HBM) and replay it")
    ap.add_argument("--profile-steps", type=int, default=3)
    return ap.parse_args()


def build_net(dtype, device):
    import torch
    import kvq_amd  # noqa: F401
    from kvq_amd import _abi
    from kvq_amd.models import VQA_Network
    from kvq_amd.utils import synth
    cfg = synth.SWIN_T_GRPB
    net = VQA_Network({"model": {"args": {"swin_tiny_grpb": {"head": {"in_channels": 768, "hidden_channels": 64}}}}})
    wts = synth.synth_swin_weights(cfg, 0, "stress")
    hw = synth.synth_vqa_head_weights(768, 64, 0, "stress")
    sd = {f"swin_tiny_grpb_backbone.{k}": torch.from_numpy(v) for k, v in wts.items()}
    sd.update({f"swin_tiny_grpb_head.{k}": torch.from_numpy(v) for k, v in hw.items()})
    net.load_state_dict(sd, strict=False)
    net.swin_tiny_grpb_backbone.operand_dtype = _abi.dtype_code(dtype)
    return net.to(device).eval(), cfg, wts, hw


def cpu_baseline(cfg, wts, hw, n_clips):
    """The oracle (CPU restatement, pinned to the reference) on this box's host cores: B=1 clip per
    forward like the reference's val loader (trainer.py:121), fp32."""
    import torch
    from kvq_amd.utils import synth
    from oracle import swin3d_oracle as O
    # 32 threads is the measured optimum on the 256-thread host (8: 3.13, 32: 2.78, 64: 3.38, 128: 8.7 s/clip)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    x = torch.from_numpy(synth.synth_clip(1000, 32, 224, 224, batch=1))
    with torch.no_grad():
        O.vqa_head(O.swin3d_trunk(x[:, :, :8, :64, :64].contiguous(), wts, cfg), hw)    # warm the allocator/threads
        t0 = time.perf_counter()
        for i in range(n_clips):
            x = torch.from_numpy(synth.synth_clip(1001 + i, 32, 224, 224, batch=1))
            O.vqa_head(O.swin3d_trunk(x, wts, cfg), hw)
        dt = time.perf_counter() - t0
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": n_clips / CLIPS_PER_VIDEO / dt, "unit": "videos/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{n_clips} clips of 3x32x224x224 (= {n_clips / CLIPS_PER_VIDEO:g} video), "
            f"B=1 per forward, fp32 torch CPU oracle, {dt:.1f} s", "cpu": model}


def main():
    args = parse()
    import torch
    import kvq_amd  # noqa: F401
    from kvq_amd import dist as kd
    from kvq_amd.utils import synth

    rank, local_rank, world = kd.init()
    assert world == args.gpus or world == 1, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # KVQ_BENCH_ONE_GPU=1 (tests): all ranks share cuda:0 so the N>1 code path can run on a 1-GPU box
    device = torch.device("cuda", 0 if os.environ.get("KVQ_BENCH_ONE_GPU") else local_rank)
    torch.cuda.set_device(device)
    net, cfg, wts, hw = build_net(args.dtype, device)
    B = args.batch
    # synthetic clips of this rank's shard, resident in HBM before the timed region
    x = torch.from_numpy(synth.synth_clip(1234 + rank, 32, 224, 224, batch=B)).to(device)
    inputs = {"technical": x}
    scores = torch.zeros(args.steps, B, device=device)
    nstream = max(1, args.streams)
    side = [torch.cuda.Stream(device=device) for _ in range(nstream - 1)]
    parts = [{"technical": t.contiguous()} for t in x.chunk(nstream)]
    by_step = args.overlap == "steps" and nstream > 1

    def forward():
        """one step = the whole batch once; with --streams > 1 the clips are split over HIP streams"""
        if nstream == 1:
            return net(inputs=inputs, reduce_scores=True).reshape(-1)
        main = torch.cuda.current_stream()
        outs = [None] * nstream
        for st in side:
            st.wait_stream(main)
        for i, st in enumerate(side):
            with torch.cuda.stream(st):
                outs[i + 1] = net(inputs=parts[i + 1], reduce_scores=True).reshape(-1)
        outs[0] = net(inputs=parts[0], reduce_scores=True).reshape(-1)
        for st in side:
            main.wait_stream(st)
        return torch.cat(outs)

    graphs = []

    def capture():
        """one hipGraph per lane: the lane's stream runs two eager steps (plans, workspaces, caches), then records a third"""
        lanes = (side + [torch.cuda.Stream(device=device)]) if by_step else [torch.cuda.Stream(device=device)]
        for cap in lanes:                                   # capture needs non-default streams
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap):
                for _ in range(2):
                    net(inputs=inputs, reduce_scores=True)
            cap.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap):
                o = net(inputs=inputs, reduce_scores=True).reshape(-1)
            graphs.append((g, o, cap))
        torch.cuda.synchronize()

    def run_graphs(n, out):
        main = torch.cuda.current_stream()
        for _, _, st in graphs:
            st.wait_stream(main)
        for s in range(n):
            g, o, st = graphs[s % len(graphs)]
            with torch.cuda.stream(st):
                g.replay()
                out[s].copy_(o, non_blocking=True)
        for _, _, st in graphs:
            main.wait_stream(st)

    def run_steps(n, out):
        if graphs:
            return run_graphs(n, out)
        if not by_step:
            for s in range(n):
                out[s] = forward()
            return
        # consecutive steps (whole batches, independent of each other) go to alternating HIP streams: every launch keeps the
        # full batch's grid, the streams fill each other's launch gaps and tails; all n steps end before the caller's sync
        main = torch.cuda.current_stream()
        lanes = [main] + side
        for st in side:
            st.wait_stream(main)
        for s in range(n):
            with torch.cuda.stream(lanes[s % nstream]):
                out[s] = net(inputs=inputs, reduce_scores=True).reshape(-1)
        for st in side:
            main.wait_stream(st)

    if by_step:      # per-stream set-up (plans + workspaces: allocations), not steps: done before the warm-up
        for st in [torch.cuda.current_stream()] + side:
            with torch.cuda.stream(st):
                net.swin_tiny_grpb_backbone.prepare(B, 32, 224, 224, device)
        torch.cuda.synchronize()
    with torch.no_grad():
        if args.graph:
            capture()
        run_steps(args.warmup, torch.zeros(max(args.warmup, 1), B, device=device))
        torch.cuda.synchronize()
        kd.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps, scores)
        # the path's one exchange step: all-gather of the per-rank score vectors (trainer_ddp.py:259-267)
        local = scores.reshape(-1)
        allscores = kd.gather_scores(local, local.numel() * world, rank, world) if world > 1 else local
        torch.cuda.synchronize()
        kd.barrier()
        torch.cuda.synchronize()
        dt = kd.max_over_ranks(time.perf_counter() - t0, device)
    clips = args.steps * B * world
    value = clips / CLIPS_PER_VIDEO / dt

    # ---- roofline of the dominant kernel: hipEvents around every launch, on the launch stream ----
    roof = None
    if rank == 0 and args.profile_steps > 0:
        bb = net.swin_tiny_grpb_backbone
        bb.profile(B, 32, 224, 224, device, True)
        with torch.no_grad():
            for _ in range(args.profile_steps):
                net(inputs=inputs, reduce_scores=True)
        recs = bb.profile_read(B, 32, 224, 224, device)
        bb.profile(B, 32, 224, 224, device, False)
        agg = {}
        for r in recs:
            a = agg.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, n=0))
            a["ms"] += r["ms"]; a["flops"] += r["flops"]; a["bytes"] += r["bytes"]; a["n"] += 1
        total_ms = sum(a["ms"] for a in agg.values())
        name, top = max(agg.items(), key=lambda kv: kv[1]["ms"])
        is_mfma = top["flops"] > 0
        if is_mfma:
            ach = top["flops"] / (top["ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / MFMA_PEAK_TFLOPS, "traffic": None}
        else:
            ach = top["bytes"] / (top["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None}
        # HBM traffic of that kernel: PMC counters from a SEPARATE rocprofv3 --pmc pass (tools/pmc_traffic.py ->
        # profiles/pmc_traffic.json; FETCH_SIZE x2 gfx950 correction), per launch like `achieved`
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"].get(name)
            if pmc:
                roof["traffic"] = pmc["fetch_bytes"] + pmc["write_bytes"]
        except (OSError, ValueError, KeyError):
            pass
        roof.update({"kernel": name, "launches_per_step": top["n"] / args.profile_steps,
                     "avg_launch_us": 1e3 * top["ms"] / top["n"],
                     "share_of_gpu_time": top["ms"] / total_ms,
                     "alg_flops_per_launch": top["flops"] / top["n"], "alg_bytes_per_launch": top["bytes"] / top["n"],
                     "step_gpu_ms": total_ms / args.profile_steps,
                     "whole_step_tflops": SWIN_T_GFLOP_PER_CLIP * B / (total_ms / args.profile_steps),
                     "by_kernel_ms_per_step": {k: round(v["ms"] / args.profile_steps, 4)
                                               for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}})
    if rank == 0:
        out = {
            "metric": "videos/sec (8-frag x 32 x 224 x 224)", "value": value, "unit": "videos/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic (PCG64 clips + procedurally generated 'stress' weights, resident in HBM)",
            "config": {"workload": "C2: KSVQE Swin3D-T(GRPB) trunk + VQAHead, 3x32x224x224 clips, video = 8 clips",
                       "clips_per_gpu_per_step": B, "operand_dtype": args.dtype, "accumulate": "fp32",
                       "sharding": f"videos[rank::{world}], one all-gather of scores at the end",
                       "streams": nstream, "overlap": args.overlap if nstream > 1 else "none",
                       "hipgraph": bool(args.graph)},
            "clips_per_s": clips / dt,
            "model_tflops": SWIN_T_GFLOP_PER_CLIP * clips / dt / 1e3,
            "score_checksum": float(allscores.double().sum().item()),
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, wts, hw, args.cpu_clips)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
