#!/usr/bin/env python
"""Benchmark of the hot path (BASELINE.json metric): videos/sec of the per-video forward on synthetic
8-fragment x 32 x 224 x 224 clips ("video" = 8 clips).

    python bench.py --gpus N --steps K --warmup W           (N>1: launched by torch.distributed.run)

Headline leg = C2 (BASELINE.json configs[1]): a "step" = one pass of the hot path over one batch of ``--batch`` clips per
GPU (default 4): K1 (``kvq_fragment_gather``: 7x7 grid of 32x32 mini-patches + normalisation) out of a
uint8 frame stack resident in HBM -> Swin3D-T(GRPB) trunk -> VQAHead.  Every step reads DIFFERENT source clips.
One JSON line on rank 0 with the contract's keys plus
  roofline      dominant kernel of the C2 step: algorithmic flops / hipEvent-measured launch time vs the dense MFMA peak
  cpu_baseline  the CPU oracle timed on this box's host cores (N = 1 only)
  no_sampler    the same steps on pre-sampled fp32 clips (K1 outside the timed region; the round-1 definition)
  parity        the headline's operands / weights on the reference-golden clips, checked in this run against the reference's stored scores
  fp16_stress   the same steps with fp16 operands on the builder's "stress" weights (rounds 1-5's headline; the product's default operand
                type, which holds the 1e-3 gate on those weights too) — and bf16_stress (bf16 on them: does not)
  c3            BASELINE configs[2]: Swin3D-T + SlowFast-R50 on the same 8 clips (1 video per step)
  c5            BASELINE configs[4]: Swin-B on 64x256x256 clips, fp16 (video = 16 clips)
(the extra legs run at N = 1; ``--legs c2`` skips them).

Timing: the K-step block (barrier + sync, K steps, sync + barrier, max over ranks) is REPEATED until about a second of steps has been
timed; ``ms_per_step`` / ``value`` are the MEDIAN block, ``repeats`` / ``ms_per_step_min`` / ``ms_per_step_max`` say how the blocks
spread (a single 20-step block is 30 ms: box-to-box and clock noise are larger than most kernel changes).
HBM traffic (``roofline.traffic``, ``whole_step_traffic``): measured in this run by two child passes per leg under
``rocprofv3 --pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE`` (``bench.py --probe LEG``: a few serial steps of the leg, nothing else), FETCH_SIZE
doubled as the gfx950 guide prescribes; ``--no-pmc`` skips them (null).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense bf16/fp16, MI355X_MICROARCH.md chip table
N_SIMDS = 1024                     # 256 CUs x 4
MFMA_CLOCK_HZ = 2.4e9              # the clock the dense peak is quoted at (2.5 PFLOP/s = 1024 SIMDs x 1024 flop/clk x 2.4 GHz)
HBM_PEAK_GBS = 8000.0
CLIPS_PER_VIDEO = 8
SWIN_T_GFLOP_PER_CLIP = 175.53   # SURVEY.md §8d (2*MAC, GEMM-only, padding as the reference pads)
SWIN_B_GFLOP_PER_CLIP = 1892.3   # Swin-B on 64x256x256
SRC_H, SRC_W = 540, 960          # post-decode frame size of the synthetic source
MEAN, STD = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)     # fusion_datasets.py:953-954


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="clips per GPU per step (C2: 4)")
    # Round 6: the headline is BASELINE configs[1] AS WRITTEN — bf16 operands — on SURVEY §8d's synthetic weights (N(0, 0.02^2) "init"), where bf16
    # holds the 1e-3 score gate (checked in the run against the reference's stored scores).  Rounds 1-5 led with fp16 operands on the
    # builder's harder "stress" weights (the fp16_stress leg).  The chip runs this step AT ITS POWER CAP (1.32 kW, rocm-smi during the timed
    # region): the shader clock settles at 2.05 GHz with fp16 MFMAs and 2.13 GHz with bf16 ones, which is the whole +4 % between the two.
    ap.add_argument("--dtype", default=os.environ.get("KVQ_BENCH_DTYPE", "bf16"), choices=["fp16", "bf16"])
    ap.add_argument("--weights", default=None, choices=["init", "stress"],
                    help="synthetic weights: init = SURVEY §8d's N(0, 0.02^2) (default with bf16), stress = the builder's large-activation set (default with fp16)")
    ap.add_argument("--no-sampler", action="store_true",
                    help="headline on pre-sampled fp32 clips (K1 outside the timed region); default: K1 inside it")
    ap.add_argument("--two-launch-sampler", action="store_true",
                    help="K1 as its own launches (kvq_fragment_gather per clip -> the fp32 batch tensor -> forward) instead of the "
                         "default: the forward reads the patch-embedding operand through the sampler (kvq_swin3d_forward_fragments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=24, help="clips the CPU oracle is timed on (24 = 3 videos)")
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams consecutive steps alternate over (independent batches fill each other's launch gaps; "
                         "measured 1 / 2 / 3 / 4 / 5 / 6 streams: 2.12 / 1.77 / 1.76 / 1.745 / 1.84 / 1.71-1.84 ms per step)")
    ap.add_argument("--cu-mask", default="none", choices=["none", "blocks", "interleave", "xcd"],
                    help="experiment: give every stream lane its own share of the CUs (hipExtStreamCreateWithCUMask)")
    ap.add_argument("--graph", type=int, default=-1,
                    help="1: every lane replays ONE recorded forward (kvq_amd/graph.py); a step still reads its own clips - the recorded "
                         "embedding launch takes their addresses from a device table (kernels.FragmentSlot).  0: eager launches.  "
                         "-1 (default): 1 on several lanes with the sampler fused into the step, else 0")
    ap.add_argument("--legs", default="all", help="comma list of extra legs at N=1: no_sampler,two_launch,fp16_stress,bf16_stress,batch8,one_stream,latency,in_mix,c3,c5,ksvqe ('all', 'c2' = none)")
    ap.add_argument("--src-pool", type=int, default=64, help="distinct uint8 source clips kept in HBM (49.8 MB each)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--min-timed-s", type=float, default=1.0, help="repeat the K-step block until this many seconds are timed")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (traffic fields = null)")
    ap.add_argument("--probe", default=None, choices=["c2", "c2mix", "c3", "c5", "ksvqe", "ksvqe96"],
                    help="internal: run --probe-steps serial steps of one leg and exit (the command the --pmc passes profile)")
    ap.add_argument("--probe-steps", type=int, default=2)
    return ap.parse_args()


GOLDEN_CASE = {"init": "t_grpb_init_32x224", "stress": "t_grpb_stress_32x224"}      # tests/golden/trunk.npz: the reference's scores for these weights


def weight_seed(kind):
    """the synthetic weights' seed = the one the reference-golden case of that kind was made with (the in-run parity check loads nothing else)"""
    import numpy as np
    gold = np.load(os.path.join(ROOT, "tests", "golden", "trunk.npz"))
    key = f"{GOLDEN_CASE[kind]}/meta"
    return int(gold[key][0]) if key in gold.files else 0


def golden_parity(net, kind, device):
    """scores of the reference-golden clips of `kind` weights with the net's current operands against the reference's stored scores"""
    import numpy as np
    import torch
    from kvq_amd.utils import synth
    gold = np.load(os.path.join(ROOT, "tests", "golden", "trunk.npz"))
    case = GOLDEN_CASE[kind]
    if f"{case}/meta" not in gold.files:
        return None
    wseed, cseed, Bg, Tg, Hg, Wg = (int(v) for v in gold[f"{case}/meta"])
    xg = torch.from_numpy(synth.synth_clip(cseed, Tg, Hg, Wg, batch=Bg)).to(device)
    with torch.no_grad():
        sg = net(inputs={"technical": xg}, reduce_scores=True).float().cpu().numpy()
    d = float(np.abs(sg - gold[f"{case}/score"]).max())
    return {"max_abs_dscore_vs_reference_golden": d, "parity_ok": bool(d <= 1e-3), "golden": f"tests/golden/trunk.npz:{case}", "weights_seed": wseed}


def build_net(dtype, device, weights="stress", seed=None):
    import torch
    import kvq_amd  # noqa: F401
    from kvq_amd import _abi
    from kvq_amd.models import VQA_Network
    from kvq_amd.utils import synth
    cfg = synth.SWIN_T_GRPB
    net = VQA_Network({"model": {"args": {"swin_tiny_grpb": {"head": {"in_channels": 768, "hidden_channels": 64}}}}})
    seed = weight_seed(weights) if seed is None else seed
    wts = synth.synth_swin_weights(cfg, seed, weights)
    hw = synth.synth_vqa_head_weights(768, 64, seed, weights)
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
    import numpy as np
    from oracle import sampler_oracle as S
    # the same workload as the GPU step: K1 (fragment sampler + normalise, fusion_datasets.py:22-107, :1017-1020) on uint8 post-decode
    # frames that are resident before the timed region, then trunk + head
    rng = np.random.default_rng(1234)
    pool = [rng.integers(0, 256, (3, 32, SRC_H, SRC_W), dtype=np.uint8) for _ in range(4)]

    def sampled_clip(i):
        g = torch.Generator().manual_seed(123400003 + i)
        rh, rw = S.draw_fragment_offsets(32, SRC_H, SRC_W, 7, 7, 32, 32, aligned=8, generator=g)
        frag = S.spatial_fragments(pool[i % len(pool)], rh, rw, 7, 7, 32, 32, aligned=8)
        return torch.from_numpy(S.normalize(frag, MEAN, STD)[None])

    x = torch.from_numpy(synth.synth_clip(1000, 32, 224, 224, batch=1))
    with torch.no_grad():
        O.vqa_head(O.swin3d_trunk(x[:, :, :8, :64, :64].contiguous(), wts, cfg), hw)    # warm the allocator/threads
        t0 = time.perf_counter()
        ts = 0.0
        for i in range(n_clips):
            t1 = time.perf_counter()
            x = sampled_clip(i)
            ts += time.perf_counter() - t1
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
            "kind": "port", "sample": f"{n_clips} clips of 3x32x224x224 (= {n_clips / CLIPS_PER_VIDEO:g} videos), "
            f"B=1 per forward, CPU oracle: fragment sampler + normalise on uint8 3x32x{SRC_H}x{SRC_W} frames (numpy, {1e3 * ts / n_clips:.1f} ms per clip) "
            f"+ fp32 torch trunk + head = the GPU step's workload, {dt:.1f} s", "cpu": model}


class Source:
    """Synthetic post-decode frames in HBM: ``n`` uint8 clips (3, 32, 540, 960) — a video's (3, 256, 540, 960) stack is 8
    consecutive clips — i.i.d. uniform bytes, plus each clip's 7 x 7 x 4 sampler origins (hoff, woff: grid origin
    ``min(H//7*i, H-32)`` + U{0..H//7-32-1}, one draw per 8-frame block, fusion_datasets.py:64-98) drawn on the host from
    ``torch.Generator().manual_seed(1234 + clip)``."""

    def __init__(self, n, device, seed):
        import torch
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        self.clips = [torch.randint(0, 256, (3, 32, SRC_H, SRC_W), dtype=torch.uint8, device=device, generator=g)
                      for _ in range(n)]
        gh = torch.tensor([min(SRC_H // 7 * i, SRC_H - 32) for i in range(7)]).view(7, 1, 1)
        gw = torch.tensor([min(SRC_W // 7 * i, SRC_W - 32) for i in range(7)]).view(1, 7, 1)
        self.hoff, self.woff = [], []
        for i in range(n):
            cg = torch.Generator().manual_seed(seed * 100003 + i)
            rh = torch.randint(SRC_H // 7 - 32, (7, 7, 4), generator=cg)
            rw = torch.randint(SRC_W // 7 - 32, (7, 7, 4), generator=cg)
            self.hoff.append((rh + gh).int().to(device))
            self.woff.append((rw + gw).int().to(device))
        self.n = n
        self._fs = {}

    def fragments(self, first, B):
        """clips first .. first+B-1 (mod pool) as a FragmentSource: K1's arguments, for the forward that reads through the sampler"""
        from kvq_amd import kernels
        key = (first % self.n, B)
        if key not in self._fs:
            idx = [(first + b) % self.n for b in range(B)]
            self._fs[key] = kernels.FragmentSource([self.clips[i] for i in idx], [self.hoff[i] for i in idx],
                                                   [self.woff[i] for i in idx], 7, 7, 32, 32, 8, mean=MEAN, std=STD)
        return self._fs[key]

    def sample_into(self, x, first):
        """K1 as its own launch: clips first .. first+B-1 (mod pool) -> the (B, 3, 32, 224, 224) fp32 batch tensor ``x``, normalised
        (``kvq_fragment_gather_batch``: one launch for the batch)."""
        self.fragments(first, x.shape[0]).materialise(out=x)


def masked_lanes(n, mode, device):
    """Experiment (round 5): one HIP stream per lane restricted to its own share of the CUs (hipExtStreamCreateWithCUMask) — spatial
    partitioning of the chip between the lanes instead of time-multiplexing.  mode: 'blocks' = lane k takes mask bits [k*256/n, (k+1)*256/n),
    'interleave' = bits with index % n == k, 'xcd' = bits with (index % 8) * n // 8 == k (whole XCDs if bits go round-robin over the XCDs)."""
    import ctypes
    import glob
    import torch
    hip = ctypes.CDLL(glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64*"))[0])
    ncu = torch.cuda.get_device_properties(device).multi_processor_count
    out = []
    for k in range(n):
        bits = [0] * ncu
        for i in range(ncu):
            if mode == "blocks":
                on = i * n // ncu == k
            elif mode == "interleave":
                on = i % n == k
            else:
                on = (i % 8) * n // 8 == k
            bits[i] = 1 if on else 0
        words = (ctypes.c_uint32 * ((ncu + 31) // 32))()
        for i, b in enumerate(bits):
            if b:
                words[i // 32] |= 1 << (i % 32)
        st = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), words)
        if rc != 0:
            sys.exit(f"bench.py: hipExtStreamCreateWithCUMask failed ({rc})")
        out.append(torch.cuda.ExternalStream(st.value, device=device))
    return out


_LANE_POOL = None


def run_lanes(lanes, n, fn):
    """consecutive steps (whole batches, independent of each other) go to alternating HIP streams; all n steps are enqueued,
    the caller synchronises.  fn(step, lane_index) enqueues one step on the CURRENT stream and returns its output.
    KVQ_LANE_THREADS=1 (experiment): one host thread per lane enqueues that lane's steps, so the lanes of a block start together
    instead of one step's enqueue time apart."""
    import torch
    global _LANE_POOL
    main = torch.cuda.current_stream()
    others = [st for st in lanes if st != main]
    for st in others:
        st.wait_stream(main)
    outs = [None] * n
    if os.environ.get("KVQ_LANE_THREADS", "0") == "1" and len(lanes) > 1 and n >= len(lanes):
        from concurrent.futures import ThreadPoolExecutor
        if _LANE_POOL is None or _LANE_POOL._max_workers < len(lanes):
            _LANE_POOL = ThreadPoolExecutor(max_workers=len(lanes))
        dev = torch.cuda.current_device()

        def lane_job(ln):
            torch.cuda.set_device(dev)                 # device, grad mode and current stream are per-thread state
            with torch.no_grad(), torch.cuda.stream(lanes[ln]):
                for s in range(ln, n, len(lanes)):
                    outs[s] = fn(s, ln)

        for f in [_LANE_POOL.submit(lane_job, ln) for ln in range(len(lanes))]:
            f.result()
    else:
        stagger = int(os.environ.get("KVQ_LANE_STAGGER_CYCLES", "0"))      # experiment: lane i starts i * stagger clock ticks late
        for s in range(n):
            with torch.cuda.stream(lanes[s % len(lanes)]):
                if stagger and 0 < s < len(lanes):
                    torch.cuda._sleep(stagger * s)
                outs[s] = fn(s, s % len(lanes))
    for st in others:
        main.wait_stream(st)
    return outs


def timed(kd, device, fn_steps, steps, warmup, finish=None, first=None, min_s=0.0):
    """W untimed steps, then the timed block — barrier + sync, EXACTLY K steps (+ ``finish``: the path's exchange step), sync +
    barrier, max over ranks — repeated until ``min_s`` seconds of blocks have been timed (the repeat count follows from the first
    block's max-over-ranks time, so every rank runs the same number).  Returns (median block seconds, outputs of the last block,
    finish's result, stats).  ``first``: index of the first TIMED step (default: behind the warm-up steps); every block runs the
    same steps — legs that compare scores time the same clips."""
    import torch
    first = warmup if first is None else first
    fn_steps(warmup, max(0, first - warmup))
    torch.cuda.synchronize()

    def block():
        kd.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = fn_steps(steps, first)
        extra = finish(outs) if finish is not None else None
        torch.cuda.synchronize()
        kd.barrier()
        torch.cuda.synchronize()
        return kd.max_over_ranks(time.perf_counter() - t0, device), outs, extra

    dt, outs, extra = block()
    dts = [dt]
    reps = int(min(200, max(1, -(-min_s // max(dt, 1e-6)))))
    for _ in range(reps - 1):
        dt, outs, extra = block()
        dts.append(dt)
    srt = sorted(dts)
    med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    stats = {"repeats": len(dts), "ms_per_step_min": 1e3 * srt[0] / steps, "ms_per_step_max": 1e3 * srt[-1] / steps,
             "timed_s": sum(dts)}
    return med, outs, extra, stats


# ---- HBM traffic, measured: child passes of this file under rocprofv3 --pmc ------------------------------------------------
ONE_TIME_KERNELS = ("bias32_build_kernel", "tail_pack", "pack_", "at::native", "Cijk_", "fill_", "copyBuffer")


def kernel_norm(name):
    """'void kvq::block_tailmm_kernel<kvq::Fp16, true, 0, true, 3>(KvqBlockTailArgs)' -> 'block_tailmm_kernel<kvq::Fp16, true, 0, true, 3>'"""
    name = name.strip()
    if name.startswith("void "):
        name = name[5:]
    if name.endswith(")") and "(" in name:
        name = name[: name.rfind("(")]
    return name[5:] if name.startswith("kvq::") else name


def kernel_base(name):
    """... -> 'block_tailmm_kernel' (the rocprof symbol of gemm_kernel carries more template arguments than the profile label)"""
    name = kernel_norm(name)
    return name.split("<")[0].split("::")[-1]


def _weights_arg():
    """an explicit --weights travels to the child runs of this file (the default follows --dtype there as here)"""
    return ["--weights", sys.argv[sys.argv.index("--weights") + 1]] if "--weights" in sys.argv[:-1] else []


def pmc_traffic(leg, steps, dtype, batch):
    """{'per_kernel': {base name: {'bytes_per_launch', 'fetch', 'write', 'launches'}}, 'step_bytes', 'steps'} or {'error': ...}."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    acc = {}
    tmp = tempfile.mkdtemp(prefix="kvq_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(var, None)
    try:
        for counter, scale in (("FETCH_SIZE", 2.0 * 1024.0), ("WRITE_SIZE", 1024.0)):      # KB; FETCH_SIZE x2 on gfx950 (MI355X guide, HBM)
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "c", "--", sys.executable, os.path.abspath(__file__),
                   "--probe", leg, "--probe-steps", str(steps), "--dtype", dtype, "--batch", str(batch)] + _weights_arg()
            if "--two-launch-sampler" in sys.argv:
                cmd.append("--two-launch-sampler")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"}
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != counter or any(k in row["Kernel_Name"] for k in ONE_TIME_KERNELS):
                    continue
                for key in {kernel_norm(row["Kernel_Name"]), kernel_base(row["Kernel_Name"])}:
                    e = acc.setdefault(key, {"fetch": 0.0, "write": 0.0, "n_fetch": 0, "n_write": 0})
                    e["fetch" if counter == "FETCH_SIZE" else "write"] += scale * float(row["Counter_Value"])
                    e["n_fetch" if counter == "FETCH_SIZE" else "n_write"] += 1
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    per = {}
    total = 0.0
    for k, e in acc.items():
        n = max(e["n_fetch"], e["n_write"], 1)
        per[k] = {"fetch": e["fetch"] / n, "write": e["write"] / n, "bytes_per_launch": (e["fetch"] + e["write"]) / n, "launches": n}
        if "<" not in k and "::" not in k:            # every launch is counted once under its base name
            total += e["fetch"] + e["write"]
    return {"per_kernel": per, "step_bytes": total / steps, "steps": steps,
            "method": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and --pmc WRITE_SIZE in separate child passes of `bench.py --probe "
                      f"{leg}` ({steps} serial steps); one-time builder / torch kernels excluded"}


def family_of(name):
    """launch family of a kernel symbol — the rows of the KVQ_SKIP ablation (profiles/r0N_skip_ablation*.txt)"""
    b = kernel_base(name)
    if b.startswith("window_attention"):
        return "attention"
    if b == "block_tailmm_kernel":
        return "tails C>=256"
    if b == "block_tail_kernel":
        return "tails C<=192"
    if b.startswith("patch_merge"):
        return "merge"
    if b.startswith("patch_embed"):
        return "embed"
    if b.startswith("gemm"):
        return "gemm"
    if b.startswith("layernorm"):
        return "layernorm"
    if b.startswith("vqa_head"):
        return "head"
    return "other"


def in_mix_evidence(args, B, nstream, ms_per_step):
    """What the launches cost IN THE TIMED REGIME: two child passes of this file under rocprofv3 —
    (1) --kernel-trace of `--probe c2mix` (the headline's steps: one recorded forward per lane replayed on `nstream` lanes; the probe marks
        the timed steps with a 60 ms idle gap): per launch family, launches per step, average duration of a launch while the other lanes'
        launches share the chip, and the sum per step (= CU-time share: the sums add up to ~nstream x the step);
    (2) --pmc SQ_VALU_MFMA_BUSY_CYCLES of `--probe c2` (the same launches, one after the other: the counter is a per-dispatch sum of
        matrix-pipe busy cycles over all SIMDs, 32 per v_mfma_f32_32x32x16 — it does not depend on what shares the chip): busy cycles per
        step / (1024 SIMDs x clock x ms_per_step of the TIMED 4-lane region) = the MFMA-busy fraction of the timed region.
    Returns a dict for roofline["in_mix"] (or {"error": ...})."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="kvq_mix_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(var, None)
    K = 24
    res = {}
    try:
        out = os.path.join(tmp, "trace")
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "--probe", "c2mix",
               "--probe-steps", str(K), "--dtype", args.dtype, "--batch", str(B), "--streams", str(nstream)] + _weights_arg()
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        files = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return {"error": f"rocprofv3 --kernel-trace failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"}
        rows = [(int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"]) for x in csv.DictReader(open(files[0]))]
        rows.sort()
        # the timed steps come behind the LAST idle gap >= 40 ms (the probe sleeps 60 ms in front of them)
        cut, last_end = 0, rows[0][1]
        for i, (st, en, _) in enumerate(rows):
            if st - last_end >= 40_000_000:
                cut = i
            last_end = max(last_end, en)
        rows = [x for x in rows[cut:] if not any(k in x[2] for k in ONE_TIME_KERNELS)]
        span_ms = (max(x[1] for x in rows) - rows[0][0]) / 1e6
        fam = {}
        for st, en, name in rows:
            f = fam.setdefault(family_of(name), {"n": 0, "us": 0.0})
            f["n"] += 1
            f["us"] += (en - st) / 1e3
        res["families"] = {k: {"launches_per_step": round(v["n"] / K, 2), "avg_launch_us_in_mix": round(v["us"] / v["n"], 2),
                               "sum_us_per_step": round(v["us"] / K, 1)} for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["us"])}
        res["traced_steps"] = K
        res["traced_ms_per_step"] = span_ms / K
        res["sum_of_launch_ms_per_step"] = sum(v["us"] for v in fam.values()) / K / 1e3
        res["avg_concurrent_launches"] = res["sum_of_launch_ms_per_step"] / res["traced_ms_per_step"]
        out = os.path.join(tmp, "pmc")
        psteps = 2
        cmd = [exe, "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "--output-format", "csv", "-d", out, "-o", "c", "--", sys.executable, os.path.abspath(__file__),
               "--probe", "c2", "--probe-steps", str(psteps), "--dtype", args.dtype, "--batch", str(B)] + _weights_arg()
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            res["mfma_busy_error"] = f"rocprofv3 --pmc failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"
            return res
        busy, byfam = 0.0, {}
        for row in csv.DictReader(open(files[0])):
            if row["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES" or any(k in row["Kernel_Name"] for k in ONE_TIME_KERNELS):
                continue
            busy += float(row["Counter_Value"])
            byfam[family_of(row["Kernel_Name"])] = byfam.get(family_of(row["Kernel_Name"]), 0.0) + float(row["Counter_Value"])
        busy /= psteps
        res["mfma_busy_cycles_per_step"] = busy
        res["mfma_busy_frac_of_timed_region"] = busy / (N_SIMDS * MFMA_CLOCK_HZ * ms_per_step * 1e-3)
        res["mfma_issued_over_algorithmic_flops"] = busy * 1024.0 / (SWIN_T_GFLOP_PER_CLIP * 1e9 * B)
        res["mfma_busy_ms_per_step_by_family"] = {k: round(v / psteps / (N_SIMDS * MFMA_CLOCK_HZ) * 1e3, 4) for k, v in sorted(byfam.items(), key=lambda kv: -kv[1])}
        res["method"] = ("families: rocprofv3 --kernel-trace of `bench.py --probe c2mix` (recorded forwards replayed on the headline's lanes); "
                         "MFMA busy: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES of `bench.py --probe c2` (per-dispatch sums over all SIMDs, 32 cycles per "
                         f"32x32x16 MFMA) / ({N_SIMDS} SIMDs x {MFMA_CLOCK_HZ / 1e9:.1f} GHz x the timed ms_per_step)")
    except Exception as e:  # noqa: BLE001
        res["error"] = f"{type(e).__name__}: {str(e)[:200]}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def attach_traffic(roof, pmc):
    """dominant kernel's measured bytes per launch + the whole step's bytes into a roofline dict"""
    if roof is None:
        return
    if not pmc or "error" in pmc:
        roof["traffic"] = None
        roof["traffic_note"] = (pmc or {}).get("error", "skipped (--no-pmc)")
        return
    row = pmc["per_kernel"].get(kernel_norm(roof["kernel"])) or pmc["per_kernel"].get(kernel_base(roof["kernel"]))
    roof["traffic"] = row["bytes_per_launch"] if row else None
    roof["traffic_detail"] = row
    roof["whole_step_traffic"] = pmc["step_bytes"]
    roof["traffic_source"] = pmc["method"]


def roofline_from_records(recs, n_steps, gflop_per_step):
    """recs: [{kernel, ms, flops, bytes}] of n_steps steps, one launch each -> the dominant kernel's roofline (algorithmic flops or
    bytes of its launches / their hipEvent-measured time) + the per-kernel table + the whole step against the MFMA peak."""
    agg = {}
    for r in recs:
        a = agg.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, n=0))
        a["ms"] += r["ms"]; a["flops"] += r["flops"]; a["bytes"] += r["bytes"]; a["n"] += 1
    total_ms = sum(a["ms"] for a in agg.values())
    name, top = max(agg.items(), key=lambda kv: kv[1]["ms"])
    if top["flops"] > 0:
        ach = top["flops"] / (top["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                "traffic": None}
    else:
        ach = top["bytes"] / (top["ms"] * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None}
    roof.update({"kernel": name, "launches_per_step": top["n"] / n_steps, "avg_launch_us": 1e3 * top["ms"] / top["n"],
                 "share_of_gpu_time": top["ms"] / total_ms, "alg_flops_per_launch": top["flops"] / top["n"],
                 "alg_bytes_per_launch": top["bytes"] / top["n"], "step_gpu_ms": total_ms / n_steps,
                 "whole_step_tflops": gflop_per_step / (total_ms / n_steps),
                 "whole_step_frac": gflop_per_step / (total_ms / n_steps) / MFMA_PEAK_TFLOPS,
                 "timing": "hipEvent brackets around every launch, on the stream the launches go to (kvq_swin3d_profile / "
                           "kvq_convnet_profile), steps run one after the other",
                 "by_kernel_ms_per_step": {k: round(v["ms"] / n_steps, 4)
                                           for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}})
    return roof


def trunk_records(bb, fwd, B, T, H, W, dev, n_steps):
    import torch
    bb.profile(B, T, H, W, dev, True)
    with torch.no_grad():
        for _ in range(n_steps):
            fwd()
    torch.cuda.synchronize()
    recs = bb.profile_read(B, T, H, W, dev)
    bb.profile(B, T, H, W, dev, False)
    return recs


def c2_roofline(net, inputs, B, profile_steps):
    bb = net.swin_tiny_grpb_backbone
    recs = trunk_records(bb, lambda: net(inputs=inputs, reduce_scores=True), B, 32, 224, 224, inputs["technical"].device, profile_steps)
    return roofline_from_records(recs, profile_steps, SWIN_T_GFLOP_PER_CLIP * B)


def setup_c3(args, device, net, src):
    """configs[2]: one video (8 clips) per step: K1 -> Swin3D-T + head on one stream, pathway packing + SlowFast-R50 (blocks
    0-4 + pools) on a second stream, both from the same sampled batch; consecutive videos alternate over two lane pairs."""
    import torch
    from kvq_amd.models.backbones.slowfast_model import slowfast
    sf = slowfast(operand_dtype=args.dtype, two_lanes=False).to(device).eval()     # the trunk already fills the chip from its stream
    B = 8
    nl = max(1, int(os.environ.get("KVQ_C3_LANES", "2")))      # lane pairs (a Swin stream + a SlowFast stream each)
    swin_st = [torch.cuda.Stream(device=device) for _ in range(nl)]
    sf_st = [torch.cuda.Stream(device=device) for _ in range(nl)]
    xs = [torch.empty(B, 3, 32, 224, 224, device=device) for _ in range(nl)]
    with torch.no_grad():
        for ln in range(nl):
            with torch.cuda.stream(swin_st[ln]):
                net.swin_tiny_grpb_backbone.prepare(B, 32, 224, 224, device)
        torch.cuda.synchronize()

    # replay (args.graph == 1, as the C2 line): both networks read the lane's static batch tensor xs[ln], so each is ONE recorded forward per
    # lane; K1 (which writes xs[ln] from this step's frames) stays an eager launch in front of them.  A failed capture = eager launches.
    recorded = None
    if args.graph == 1:
        try:
            recorded = []
            with torch.no_grad():
                for ln in range(nl):
                    src.sample_into(xs[ln], ln * B)
                    torch.cuda.synchronize()
                    pair = []
                    for st, fn in ((swin_st[ln], lambda ln=ln: (net(inputs={"technical": xs[ln]}, reduce_scores=True),)),
                                   (sf_st[ln], lambda ln=ln: sf.forward_clips(xs[ln]))):
                        with torch.cuda.stream(st):
                            for _ in range(2):
                                fn()
                        st.synchronize()
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=st):
                            o = fn()
                        pair.append((g, o))
                    recorded.append(pair)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            import warnings
            warnings.warn(f"C3: hipGraph capture failed ({type(e).__name__}: {str(e).splitlines()[0][:160]}); eager launches", RuntimeWarning)
            torch.cuda.synchronize()
            recorded = None

    def steps(n, first):
        main = torch.cuda.current_stream()
        for st in swin_st + sf_st:
            st.wait_stream(main)
        outs = []
        for s in range(n):
            ln = s % nl
            with torch.cuda.stream(swin_st[ln]):
                swin_st[ln].wait_stream(sf_st[ln])               # the lane's previous SlowFast pass still reads xs[ln]
                src.sample_into(xs[ln], (first + s) * B)
                sampled = torch.cuda.Event()
                sampled.record()
                if recorded is not None:
                    recorded[ln][0][0].replay()
                    score = recorded[ln][0][1][0].clone()
                else:
                    score = net(inputs={"technical": xs[ln]}, reduce_scores=True)
            with torch.cuda.stream(sf_st[ln]):
                sf_st[ln].wait_event(sampled)
                if recorded is not None:
                    recorded[ln][1][0].replay()
                    slow_f, fast_f = (t.clone() for t in recorded[ln][1][1])
                else:
                    slow_f, fast_f = sf.forward_clips(xs[ln])
            outs.append((score, slow_f, fast_f))
        for st in swin_st + sf_st:
            main.wait_stream(st)
        return outs

    def serial(n):      # the probe: the same launches, one after the other on the current stream
        for s in range(n):
            src.sample_into(xs[0], s * B)
            net(inputs={"technical": xs[0]}, reduce_scores=True)
            sf.forward_clips(xs[0])

    steps.replay = recorded is not None
    return sf, B, xs, steps, serial


def leg_c3(args, device, net, src, kd, pmc):
    import torch
    from kvq_amd.models.backbones.slowfast_model import conv_flops
    sf, B, xs, steps, _ = setup_c3(args, device, net, src)
    k = max(2, min(args.steps, 20))
    with torch.no_grad():
        dt, outs, _, stats = timed(kd, device, steps, k, max(2, min(args.warmup, 5)), min_s=args.min_timed_s)
    sf_flops, _ = conv_flops(32, 224, 224)
    flops = B * (SWIN_T_GFLOP_PER_CLIP * 1e9 + sf_flops)
    ach = flops * k / dt / 1e12
    finite = all(bool(torch.isfinite(o[0]).all() and torch.isfinite(o[1]).all() and torch.isfinite(o[2]).all()) for o in outs[-2:])
    # per-kernel view of one video: the trunk's launches (hipEvent brackets) + every op of the SlowFast plan (kvq_convnet_profile)
    roof = None
    if args.profile_steps > 0:
        with torch.no_grad():
            src.sample_into(xs[0], 0)
            recs = trunk_records(net.swin_tiny_grpb_backbone, lambda: net(inputs={"technical": xs[0]}, reduce_scores=True), B, 32, 224, 224,
                                 device, 1)
            for r in sf.profile_layers(xs[0]):
                recs.append({"kernel": "slowfast:" + r["kind"] + (f" {r['M']}x{r['N']}x{r['K']}" if r["M"] else " " + r["name"][-24:]),
                             "ms": r["ms"], "flops": r["tflops"] * 1e12 * r["ms"] * 1e-3, "bytes": 0.0})
        roof = roofline_from_records(recs, 1, flops / 1e9)
        attach_traffic(roof, pmc)
    out = {"workload": "C3: K1 + Swin3D-T(GRPB) trunk + VQAHead and SlowFast-R50 (blocks 0-4 + pools) on the same 8 clips, "
           "1 video per step, two branches on two HIP streams", "value": k / dt, "unit": "videos/s", "steps": k,
           "ms_per_step": 1e3 * dt / k, "clips_per_step": B, "dtype": args.dtype, "finite": finite, "hipgraph": bool(getattr(steps, "replay", False)),
           "alg_gflop_per_clip": {"swin3d_t": SWIN_T_GFLOP_PER_CLIP, "slowfast_r50": sf_flops / 1e9,
                                  "slowfast_counted_from": "kvq_amd.models.backbones.slowfast_model.conv_flops (2*MAC of every Conv3d "
                                  "of the restated pytorchvideo R50 8x8, the shapes oracle/slowfast_oracle.py runs)"},
           "whole_step": {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                          "scope": "whole step (both branches), wall time"},
           "roofline": roof}
    out.update(stats)
    return out


KSVQE_GFLOP_PER_SAMPLE = 367.8          # one 32-frame KSVQE sample (CLIP ViT-B/16 on 16 key frames + QRS + CONTRIQUE + trunk + CDM)


def setup_ksvqe(args, device, B, T):
    """The f1 path (KSVQE_model.py:1389-1500): CLIP visual tower + QRS + CONTRIQUE + the Swin3D-T(GRPB) trunk with CDM modulation + head,
    B samples of T frames per forward, random-init weights of the architecture, synthetic inputs of the dataset's shapes.  Forwards of
    consecutive batches replay recorded hipGraphs on 4 lanes, as Trainer._score_all runs the model (kvq_amd/graph.py)."""
    import torch
    from kvq_amd.graph import LaneGraphs
    from kvq_amd.models import VQA_Network
    from kvq_amd.utils import synth
    cfg = {"model": {"type": "KSVQE", "args": {"KSVQE": {"backbone": dict(checkpoint=True, pretrained=None, num_samples=1, sample_type="topkpertubation",
           CLIP_location=8, cls_use=True, tuning_stage=2, qls_swin=True, frozen3D=False, frozen_stages=-1),
           "head": {"in_channels": 768, "hidden_channels": 64}}}}}
    net = VQA_Network(cfg)
    sd = {"KSVQE_backbone." + k: torch.from_numpy(v) for k, v in synth.synth_ksvqe_weights(3).items()}
    sd.update({"KSVQE_head." + k: torch.from_numpy(v) for k, v in synth.synth_vqa_head_weights(768, 64, 3, "stress").items()})
    net.load_state_dict(sd, strict=False)
    net = net.to(device).eval()
    net.KSVQE_backbone.aux_loss = False               # the harness discards the loss (Trainer._score_all)
    pool = [{k: torch.from_numpy(v).to(device) for k, v in synth.synth_ksvqe_inputs(11 + i, B, T).items()} for i in range(4)]

    def fwd(inputs):
        with torch.no_grad():
            return net(inputs=dict(inputs), reduce_scores=True)[0]

    lanes = [torch.cuda.Stream(device=device) for _ in range(4)]
    graphs = LaneGraphs(fwd, lanes)

    def steps(n, first):
        main = torch.cuda.current_stream()
        for st in lanes:
            st.wait_stream(main)
        outs = []
        for s in range(n):
            ln = s % len(lanes)
            o = graphs.run(ln, pool[(first + s) % len(pool)])
            with torch.cuda.stream(lanes[ln]):
                outs.append(o.clone())
        for st in lanes:
            main.wait_stream(st)
        return outs

    def serial(n):
        for s in range(n):
            fwd(pool[s % len(pool)])

    return net, steps, serial, graphs


def leg_ksvqe(args, device, kd, pmc, T):
    import torch
    B = 4 if T == 32 else 1
    net, steps, serial, graphs = setup_ksvqe(args, device, B, T)
    k = max(8, args.steps)
    dt, outs, _, stats = timed(kd, device, steps, k, 8, min_s=args.min_timed_s)
    gf = KSVQE_GFLOP_PER_SAMPLE * (T / 32.0)          # every part of the forward is linear in the frame count at fixed resolution
    ach = gf * B * k / dt / 1e3
    out = {"workload": f"KSVQE (f1): CLIP visual tower + QRS + CONTRIQUE + Swin3D-T(GRPB) trunk + CDM + head, {B} sample(s) of {T} frames per "
                       "forward, hipGraph replay on 4 lanes (Trainer._score_all's path), random-init weights, synthetic inputs",
           "value": B * k / dt, "unit": "samples/s", "steps": k, "ms_per_step": 1e3 * dt / k, "samples_per_step": B, "frames": T,
           "dtype": "fp16", "finite": bool(torch.isfinite(torch.cat([o.reshape(-1) for o in outs])).all()),
           "graph_replays": graphs.replays, "eager_runs": graphs.eager_runs,
           "alg_gflop_per_sample": gf,
           "whole_step": {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                          "scope": "whole forward (all five parts), wall time over 4 lanes"}}
    if pmc is not None:
        out["whole_step_traffic"] = pmc.get("step_bytes") if "error" not in pmc else None
        out["traffic_note"] = pmc.get("method") or pmc.get("error")
        if "per_kernel" in pmc:       # the heaviest kernels by measured HBM bytes per forward
            top = sorted(((v["bytes_per_launch"] * v["launches"], k_) for k_, v in pmc["per_kernel"].items() if "<" not in k_ and "::" not in k_), reverse=True)[:6]
            out["traffic_by_kernel_MB_per_forward"] = {k_: round(b / pmc["steps"] / 1e6, 1) for b, k_ in top}
    out.update(stats)
    del net, graphs
    return out


def setup_c5(args, device):
    """configs[4]: Swin-B (E=128, depths 2/2/18/2, heads 4/8/16/32) on 64x256x256 clips, fp16 operands; video = 16 clips.  K1 is in
    the step: every clip is gathered as an 8 x 8 grid of 32 x 32 patches (per-8-frame offsets) out of a uint8 3x64x540x960 source."""
    import torch
    from kvq_amd import _abi, kernels
    from kvq_amd.models.backbones.swin_backbone import SwinTransformer3D
    from kvq_amd.models.head import VQAHead
    from kvq_amd.utils import synth
    cfg = synth.SWIN_B_GRPB
    bb = SwinTransformer3D(embed_dim=128, depths=list(cfg.depths), num_heads=list(cfg.num_heads)).to(device).eval()
    bb.operand_dtype = _abi.dtype_code("fp16")
    head = VQAHead(in_channels=1024, hidden_channels=64).to(device).eval()
    B = 4
    g = torch.Generator(device=device)
    g.manual_seed(77)
    npool = 8                                                                                   # 8 x 99.5 MB of uint8 frames: distinct per step pair
    clips = [torch.randint(0, 256, (3, 64, SRC_H, SRC_W), dtype=torch.uint8, device=device, generator=g) for _ in range(npool)]
    gh = torch.tensor([min(SRC_H // 8 * i, SRC_H - 32) for i in range(8)]).view(8, 1, 1)
    gw = torch.tensor([min(SRC_W // 8 * i, SRC_W - 32) for i in range(8)]).view(1, 8, 1)
    hoff, woff = [], []
    for i in range(npool):
        cg = torch.Generator().manual_seed(7700 + i)
        hoff.append((torch.randint(SRC_H // 8 - 32, (8, 8, 8), generator=cg) + gh).int().to(device))
        woff.append((torch.randint(SRC_W // 8 - 32, (8, 8, 8), generator=cg) + gw).int().to(device))
    # KVQ_C5_LANES (default 4; 2 in rounds 2-4: 23.2 -> 23.4 videos/s): stream lanes of the C5 leg (each owns a plan + 5.5 GiB workspace)
    n_lanes = max(1, int(os.environ.get("KVQ_C5_LANES", "4")))
    # replay (as the C2 line: one recorded forward per lane, the clips' addresses through a kernels.FragmentSlot) unless the sampler runs as
    # its own launches or eager launches were asked for
    replay = args.graph == 1 and not args.two_launch_sampler and n_lanes > 1          # (-1 = unresolved: the --probe path, eager)
    lanes = ([torch.cuda.Stream(device=device) for _ in range(n_lanes)] if replay else
             [torch.cuda.current_stream()] + [torch.cuda.Stream(device=device) for _ in range(n_lanes - 1)])
    xs = [torch.empty(B, 3, 64, 256, 256, device=device) if (args.two_launch_sampler or i == 0) else None for i in range(n_lanes)]
    with torch.no_grad():
        for st in lanes:
            with torch.cuda.stream(st):
                bb.prepare(B, 64, 256, 256, device)
        torch.cuda.synchronize()

    frs = {}

    def one(s, ln):
        idx = [(s * B + b) % npool for b in range(B)]
        if not args.two_launch_sampler:          # the embedding launch reads through the sampler
            if idx[0] not in frs:
                frs[idx[0]] = kernels.FragmentSource([clips[i] for i in idx], [hoff[i] for i in idx], [woff[i] for i in idx],
                                                     8, 8, 32, 32, 8, mean=MEAN, std=STD)
            return head(bb({"technical": frs[idx[0]]}))
        for b, i in enumerate(idx):
            kernels.fragment_gather(clips[i], hoff[i], woff[i], 8, 8, 32, 32, 8, MEAN, STD, out=xs[ln][b])
        return head(bb({"technical": xs[ln]}))

    graphs = None
    if replay:
        from kvq_amd.graph import LaneGraphs
        graphs = LaneGraphs(lambda inp: head(bb(inp)), lanes)

        def source(s):
            idx = [(s * B + b) % npool for b in range(B)]
            if idx[0] not in frs:
                frs[idx[0]] = kernels.FragmentSource([clips[i] for i in idx], [hoff[i] for i in idx], [woff[i] for i in idx],
                                                     8, 8, 32, 32, 8, mean=MEAN, std=STD)
            return {"technical": frs[idx[0]]}

        with torch.no_grad():
            for ln in range(n_lanes):
                graphs.run(ln, source(ln))
            torch.cuda.synchronize()
        if graphs.eager_runs:
            graphs = None

    def steps(n, first):
        if graphs is None:
            return run_lanes(lanes, n, lambda s, ln: one(first + s, ln))
        main = torch.cuda.current_stream()
        for st in lanes:
            st.wait_stream(main)
        outs = []
        for s in range(n):
            o = graphs.run(s % n_lanes, source(first + s))
            with torch.cuda.stream(lanes[s % n_lanes]):
                outs.append(o.clone())
        for st in lanes:
            main.wait_stream(st)
        return outs

    def serial(n):
        for s in range(n):
            one(s, 0)

    steps.replay = graphs is not None
    return bb, head, B, xs, steps, serial


def leg_c5(args, device, kd, pmc):
    import torch
    bb, head, B, xs, steps, _ = setup_c5(args, device)
    k = max(2, min(args.steps, 10))
    with torch.no_grad():
        dt, outs, _, stats = timed(kd, device, steps, k, 2, min_s=args.min_timed_s)
    ach = SWIN_B_GFLOP_PER_CLIP * B * k / dt / 1e3
    roof = None
    if args.profile_steps > 0:
        recs = trunk_records(bb, lambda: head(bb({"technical": xs[0]})), B, 64, 256, 256, device, 1)
        roof = roofline_from_records(recs, 1, SWIN_B_GFLOP_PER_CLIP * B)
        attach_traffic(roof, pmc)
    out = {"workload": "C5: K1 (8x8 grid of 32x32 patches out of uint8 3x64x540x960) + Swin-B(GRPB) trunk + VQAHead, 3x64x256x256 clips, "
           "fp16 operands, video = 16 clips", "value": B * k / 16.0 / dt,
           "unit": "videos/s", "steps": k, "ms_per_step": 1e3 * dt / k, "clips_per_step": B, "dtype": "fp16", "hipgraph": bool(getattr(steps, "replay", False)),
           "finite": bool(torch.isfinite(torch.cat([o.reshape(-1) for o in outs])).all()),
           "alg_gflop_per_clip": SWIN_B_GFLOP_PER_CLIP,
           "whole_step": {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                          "scope": "whole step, wall time"},
           "roofline": roof}
    out.update(stats)
    return out


def run_probe(args):
    """`bench.py --probe LEG`: what the rocprofv3 --pmc child passes profile — set the leg up, run --probe-steps steps of it one
    after the other on one stream, exit.  No timing, no JSON."""
    import torch
    import kvq_amd  # noqa: F401
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    with torch.no_grad():
        if args.probe == "c5":
            *_, serial = setup_c5(args, device)
            serial(args.probe_steps)
        elif args.probe in ("ksvqe", "ksvqe96"):
            T = 32 if args.probe == "ksvqe" else 96
            _, _, serial, _ = setup_ksvqe(args, device, 4 if T == 32 else 1, T)
            serial(args.probe_steps)       # (the one-time builder kernels of the first forward are excluded by name, ONE_TIME_KERNELS)
        else:
            net, *_ = build_net(args.dtype, device, args.weights or ("init" if args.dtype == "bf16" else "stress"))
            B = args.batch if args.probe in ("c2", "c2mix") else 8
            src = Source(max(B, 2 * B) if args.probe != "c2mix" else 16 * B, device, 1234)
            if args.probe == "c2mix":
                # the headline's timed regime for the kernel-trace child pass: one recorded forward per lane, replayed; the traced steps
                # come behind a 60 ms idle gap (in_mix_evidence cuts there)
                import time as _t
                from kvq_amd.graph import LaneGraphs
                n = max(1, args.streams)
                glanes = [torch.cuda.Stream(device=device) for _ in range(n)]
                for st in glanes:
                    with torch.cuda.stream(st):
                        net.swin_tiny_grpb_backbone.prepare(B, 32, 224, 224, device)
                torch.cuda.synchronize()
                graphs = LaneGraphs(lambda inp: net(inputs=inp, reduce_scores=True), glanes)
                for s in range(2 * n):
                    graphs.run(s % n, {"technical": src.fragments(s * B, B)})
                torch.cuda.synchronize()
                assert graphs.eager_runs == 0, "the c2mix probe must replay"
                _t.sleep(0.06)
                for s in range(args.probe_steps):
                    graphs.run(s % n, {"technical": src.fragments(s * B, B)})
                torch.cuda.synchronize()
            elif args.probe == "c2":
                net.swin_tiny_grpb_backbone.prepare(B, 32, 224, 224, device)
                x = torch.empty(B, 3, 32, 224, 224, device=device)
                for s in range(args.probe_steps):
                    if args.two_launch_sampler:
                        src.sample_into(x, s * B)
                        net(inputs={"technical": x}, reduce_scores=True)
                    else:
                        net(inputs={"technical": src.fragments(s * B, B)}, reduce_scores=True)
            else:
                *_, serial = setup_c3(args, device, net, src)
                serial(args.probe_steps)
    torch.cuda.synchronize()


def relaunch_ranks(args):
    """``python bench.py --gpus N`` without a rank environment: become ``torch.distributed.run`` with N ranks on this node (one per
    GPU, RCCL over xGMI) — a single process must never report itself as an N-GPU run (nor silently fall back to one GPU)."""
    port = os.environ.get("MASTER_PORT") or str(20000 + (os.getpid() * 7919) % 20000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    args = parse()
    if args.probe:
        return run_probe(args)
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_ranks(args)
    import torch
    import kvq_amd  # noqa: F401
    from kvq_amd import _abi, dist as kd
    from kvq_amd.utils import synth

    rank, local_rank, world = kd.init()
    if world != args.gpus:
        sys.exit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # KVQ_BENCH_ONE_GPU=1 (tests): all ranks share cuda:0 so the N>1 code path can run on a 1-GPU box
    device = torch.device("cuda", 0 if os.environ.get("KVQ_BENCH_ONE_GPU") else local_rank)
    torch.cuda.set_device(device)
    weights_kind = args.weights or ("init" if args.dtype == "bf16" else "stress")
    net, cfg, wts, hw = build_net(args.dtype, device, weights_kind)
    bb = net.swin_tiny_grpb_backbone
    headline_parity = golden_parity(net, weights_kind, device) if rank == 0 else None       # before anything is timed
    B = args.batch
    nstream = max(1, args.streams)
    lanes = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=device) for _ in range(nstream - 1)]
    if args.cu_mask != "none":
        lanes = masked_lanes(nstream, args.cu_mask, device)
    legs = {"no_sampler", "two_launch", "fp16_stress", "bf16_stress", "batch8", "one_stream", "latency", "in_mix", "c3", "c5", "ksvqe"} if args.legs == "all" else {x for x in args.legs.split(",") if x and x != "c2"}
    if world > 1:
        legs = set()

    # ---- synthetic source of this rank's shard, resident in HBM before the timed region --------------------------------
    need = (args.warmup + args.steps) * B
    src = Source(min(need, max(B, args.src_pool)), device, 1234 + rank)
    xs = [torch.empty(B, 3, 32, 224, 224, device=device) for _ in range(nstream)]      # one batch tensor per lane
    with torch.no_grad():
        for st in lanes:      # per-stream set-up (plans + workspaces: allocations), not steps: done before the warm-up
            with torch.cuda.stream(st):
                bb.prepare(B, 32, 224, 224, device)
        torch.cuda.synchronize()

    def step_two_launch(s, ln):
        src.sample_into(xs[ln], s * B)
        return net(inputs={"technical": xs[ln]}, reduce_scores=True)

    def step_fused(s, ln):
        return net(inputs={"technical": src.fragments(s * B, B)}, reduce_scores=True)

    step_sampled = step_two_launch if args.two_launch_sampler else step_fused

    # pre-sampled clips for the --no-sampler definition: K1 run once, outside the timed region, distinct per step
    pre = None
    if args.no_sampler or "no_sampler" in legs:
        npre = min(need, src.n) // B
        pre = [torch.empty(B, 3, 32, 224, 224, device=device) for _ in range(max(1, npre))]
        for i, t in enumerate(pre):
            src.sample_into(t, i * B)
        torch.cuda.synchronize()

    def step_presampled(s, ln):
        return net(inputs={"technical": pre[s % len(pre)]}, reduce_scores=True)

    sampler_on = not args.no_sampler
    # --graph 1 (below): every lane replays ONE recorded forward (kvq_amd/graph.py, the harness's own replay path); a step still reads its own
    # clips — the recorded embedding launch takes the frames' and draws' addresses from a 384-byte device table
    # (kernels.FragmentSlot / KvqFragmentSource.indirect) that is rewritten on the lane's stream in front of the replay
    glanes = None
    graphable = {}            # step function -> the input of step s, for the definitions a recorded forward can serve
    graph_state = {"used": False, "fell_back": False}

    def steps_of(step_fn):
        """run(n, first): enqueue steps first .. first+n-1 of this definition over the lanes (eager launches; --graph 1: replays of a
        forward recorded HERE, i.e. with the weights / operand type in force when the leg starts)"""
        if args.graph and step_fn in graphable:
            from kvq_amd.graph import LaneGraphs
            graph_input = graphable[step_fn]
            graphs = LaneGraphs(lambda inp: net(inputs=inp, reduce_scores=True), glanes)
            with torch.no_grad():
                for ln in range(nstream):          # record every lane before anything is timed
                    graphs.run(ln, graph_input(ln))
                torch.cuda.synchronize()
            if graphs.eager_runs:                  # a capture failed (LaneGraphs warned): this definition runs with eager launches
                graph_state["fell_back"] = True
                return lambda n, first: run_lanes(lanes, n, lambda s, ln: step_fn(first + s, ln))
            graph_state["used"] = True

            def run(n, first):
                main = torch.cuda.current_stream()
                for st in glanes:
                    st.wait_stream(main)
                outs = []
                for s in range(n):
                    ln = s % nstream
                    o = graphs.run(ln, graph_input(first + s))
                    with torch.cuda.stream(glanes[ln]):
                        outs.append(o.clone())            # the graph's static output is overwritten by the lane's next replay
                for st in glanes:
                    main.wait_stream(st)
                return outs
            return run
        return lambda n, first: run_lanes(lanes, n, lambda s, ln: step_fn(first + s, ln))

    def finish(outs):
        # the path's one exchange step: all-gather of the per-rank score vectors (trainer_ddp.py:259-267)
        local = torch.cat([o.reshape(-1) for o in outs])
        return kd.gather_scores(local, local.numel() * world, rank, world) if world > 1 else local

    if args.graph < 0:
        args.graph = int(nstream > 1 and sampler_on and not args.two_launch_sampler and args.cu_mask == "none")
    if args.graph and glanes is None:
        prio = [int(v) for v in os.environ.get("KVQ_LANE_PRIO", "").split(",") if v.strip()]       # experiment: per-lane stream priority
        glanes = [torch.cuda.Stream(device=device, priority=prio[i % len(prio)] if prio else 0) for i in range(nstream)]
    graphable[step_fused] = lambda s: {"technical": src.fragments(s * B, B)}       # pre-sampled / two-launch definitions stay eager
    with torch.no_grad():
        dt, outs, allscores, tstats = timed(kd, device, steps_of(step_sampled if sampler_on else step_presampled), args.steps,
                                            args.warmup, finish, min_s=args.min_timed_s)
    headline_graph = graph_state["used"] and not graph_state["fell_back"]
    clips = args.steps * B * world
    value = clips / CLIPS_PER_VIDEO / dt
    fp_scores = torch.cat([o.reshape(-1) for o in outs]).float().cpu()

    out = None
    if rank == 0:
        roof = None
        want_pmc = world == 1 and not args.no_pmc and args.profile_steps > 0
        if args.profile_steps > 0:
            src.sample_into(xs[0], 0)
            fused = sampler_on and not args.two_launch_sampler
            roof = c2_roofline(net, {"technical": src.fragments(0, B) if fused else xs[0]}, B, args.profile_steps)
            attach_traffic(roof, pmc_traffic("c2", 2, args.dtype, B) if want_pmc else None)
        out = {
            "metric": "videos/sec (8-frag x 32 x 224 x 224)", "value": value, "unit": "videos/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic (uint8 frames i.i.d. uniform, seeded sampler offsets, "
                    + ("random-init weights N(0, 0.02^2) as SURVEY §8d specifies" if weights_kind == "init" else "procedurally generated 'stress' weights")
                    + "; all resident in HBM before the timed region)",
            "config": {"workload": "C2: KSVQE Swin3D-T(GRPB) trunk + VQAHead, 3x32x224x224 clips, video = 8 clips"
                                   + (", fragment sampler K1 (uint8 3x32x540x960 per clip -> 7x7 grid of 32x32 patches, normalised) "
                                      "inside the step" + (" as its own launches" if args.two_launch_sampler else
                                                           ", read through by the patch-embedding launch (no fp32 clip in HBM)")
                                      if sampler_on else ", pre-sampled fp32 clips (K1 outside the step)"),
                       "clips_per_gpu_per_step": B, "operand_dtype": args.dtype, "accumulate": "fp32", "weights": weights_kind,
                       "parity": headline_parity,
                       "sampler_in_step": sampler_on,
                       "sampler": ("kvq_fragment_gather_batch (one launch per batch), then the forward" if args.two_launch_sampler or not sampler_on
                                   else "kvq_swin3d_forward_fragments (K1 fused into the embedding's operand read; bit-identical scores)"),
                       "source_pool_clips": src.n, "distinct_clips_per_step": True,
                       "sharding": f"videos[rank::{world}], one all-gather of scores at the end",
                       "streams": nstream, "overlap": "steps" if nstream > 1 else "none",
                       "hipgraph": bool(headline_graph),
                       "launches": ("one recorded forward per lane, replayed (kvq_amd/graph.py LaneGraphs, the harness's path for lazy samples); "
                                    "each step's frame / draw addresses reach the recorded embedding launch through a 384-byte device table "
                                    "(kernels.FragmentSlot) rewritten on the lane's stream in front of the replay"
                                    if headline_graph else "eager, one C call per step")},
            "repeats": tstats["repeats"], "ms_per_step_min": tstats["ms_per_step_min"], "ms_per_step_max": tstats["ms_per_step_max"],
            "timed_s": tstats["timed_s"], "timing": "median of `repeats` blocks of exactly `steps` steps, each bracketed by barrier + "
                                                    "synchronize and reduced to the max over ranks",
            "clips_per_s": clips / dt,
            "model_tflops": SWIN_T_GFLOP_PER_CLIP * clips / dt / 1e3,
            "whole_job_frac_of_mfma_peak": SWIN_T_GFLOP_PER_CLIP * clips / dt / 1e3 / MFMA_PEAK_TFLOPS / world,
            "score_checksum": float(allscores.double().sum().item()),
            "roofline": roof,
        }
    # ---- extra legs (N = 1): the other definitions / configs, each timed the same way -----------------------------------
    if rank == 0 and world == 1:
        with torch.no_grad():
            if "no_sampler" in legs and sampler_on:
                dt2, _, _, st2 = timed(kd, device, steps_of(step_presampled), args.steps, min(args.warmup, 5), min_s=args.min_timed_s)
                out["no_sampler"] = {"value": args.steps * B / CLIPS_PER_VIDEO / dt2, "unit": "videos/s",
                                     "ms_per_step": 1e3 * dt2 / args.steps, "steps": args.steps, "repeats": st2["repeats"],
                                     "note": "same steps on pre-sampled fp32 clips (K1 outside the timed region, distinct clips per "
                                             "step): the round-1 definition of the step; eager launches"}
            if "two_launch" in legs and sampler_on and not args.two_launch_sampler:
                dt4, outs4, _, st4 = timed(kd, device, steps_of(step_two_launch), args.steps, min(args.warmup, 5), first=args.warmup,
                                           min_s=args.min_timed_s)
                s4 = torch.cat([o.reshape(-1) for o in outs4]).float().cpu()
                out["two_launch_sampler"] = {"value": args.steps * B / CLIPS_PER_VIDEO / dt4, "unit": "videos/s",
                                             "ms_per_step": 1e3 * dt4 / args.steps, "steps": args.steps, "repeats": st4["repeats"],
                                             "scores_equal_headline": bool(torch.equal(s4, fp_scores)),
                                             "note": "the same steps with K1 as its own launch (kvq_fragment_gather_batch writes the fp32 batch tensor, the "
                                                     "forward reads it back): rounds 2-4's step (which launched K1 once per clip)"}
            # the other operand type / weight set pairs, same steps, lanes and sampler: fp16 on the "stress" weights = rounds 1-5's headline and the
            # product's default operands (holds the 1e-3 gate there too); bf16 on them does not (8-bit mantissa)
            def swap(dtype, kind):
                seed = weight_seed(kind)
                sd_ = {f"swin_tiny_grpb_backbone.{k}": torch.from_numpy(v) for k, v in synth.synth_swin_weights(cfg, seed, kind).items()}
                sd_.update({f"swin_tiny_grpb_head.{k}": torch.from_numpy(v) for k, v in synth.synth_vqa_head_weights(768, 64, seed, kind).items()})
                net.load_state_dict(sd_, strict=False)
                bb.operand_dtype = _abi.dtype_code(dtype)
                for st in lanes:
                    with torch.cuda.stream(st):
                        bb.prepare(B, 32, 224, 224, device)
                torch.cuda.synchronize()
            swapped = False
            for name, dt_, kind in (("fp16_stress", "fp16", "stress"), ("bf16_stress", "bf16", "stress")):
                if name not in legs or (dt_, kind) == (args.dtype, weights_kind):
                    continue
                swap(dt_, kind)
                swapped = True
                par = golden_parity(net, kind, device)
                dtx, _, _, stx = timed(kd, device, steps_of(step_sampled if sampler_on else step_presampled), args.steps,
                                       min(args.warmup, 5), first=args.warmup, min_s=args.min_timed_s)
                out[name] = {"value": args.steps * B / CLIPS_PER_VIDEO / dtx, "unit": "videos/s", "ms_per_step": 1e3 * dtx / args.steps,
                             "steps": args.steps, "repeats": stx["repeats"], "operand_dtype": dt_, "weights": kind, **(par or {}),
                             "note": ("rounds 1-5's headline: fp16 operands (the product's default) on the builder's 'stress' weights"
                                      if name == "fp16_stress" else "bf16 operands on the 'stress' weights: the 8-bit mantissa does not hold the "
                                      "1e-3 gate there (tests/test_gpu_e2e.py, DESIGN.md §2)")}
            if swapped:
                swap(args.dtype, weights_kind)
            if "batch8" in legs and sampler_on:
                # information only: the same step at one whole video (8 clips) per step.  The headline stays at BASELINE configs[1]'s batch = 4.
                B8 = 2 * B
                xs8 = [torch.empty(B8, 3, 32, 224, 224, device=device) for _ in range(nstream)]
                for st in lanes:
                    with torch.cuda.stream(st):
                        bb.prepare(B8, 32, 224, 224, device)
                torch.cuda.synchronize()

                def step8(s, ln):
                    if not args.two_launch_sampler:
                        return net(inputs={"technical": src.fragments(s * B8, B8)}, reduce_scores=True)
                    src.sample_into(xs8[ln], s * B8)
                    return net(inputs={"technical": xs8[ln]}, reduce_scores=True)
                k8 = max(4, args.steps // 2)
                if not args.two_launch_sampler:
                    graphable[step8] = lambda s: {"technical": src.fragments(s * B8, B8)}
                dt8, _, _, st8 = timed(kd, device, steps_of(step8), k8,
                                       min(args.warmup, 4), min_s=args.min_timed_s)
                out["batch8"] = {"value": k8 * B8 / CLIPS_PER_VIDEO / dt8, "unit": "videos/s", "ms_per_step": 1e3 * dt8 / k8, "steps": k8,
                                 "repeats": st8["repeats"], "clips_per_step": B8,
                                 "note": "the headline's step at 8 clips (one video) per step on the same stream lanes - not BASELINE configs[1]'s "
                                         "batch = 4, reported beside it"}
                del xs8
                torch.cuda.empty_cache()
            if "one_stream" in legs and sampler_on and nstream > 1:
                # the same steps, the same (default) launch geometries, eager launches on ONE stream: what the multi-lane geometries of
                # rounds 5-6 cost a deployment that runs one video at a time (the launches are shaped for CU x time, not for latency)
                k1 = max(4, args.steps // 2)
                one = lambda n, first: run_lanes(lanes[:1], n, lambda s_, ln: step_sampled(first + s_, 0))      # noqa: E731
                dt6, _, _, st6 = timed(kd, device, one, k1, min(args.warmup, 3), min_s=min(args.min_timed_s, 0.5))
                out["one_stream"] = {"value": k1 * B / CLIPS_PER_VIDEO / dt6, "unit": "videos/s", "ms_per_step": 1e3 * dt6 / k1, "steps": k1,
                                     "repeats": st6["repeats"], "note": "the headline's steps one after the other on one stream (eager launches, "
                                     "the default multi-lane launch geometries)"}
            if "latency" in legs and sampler_on:
                # KVQ_LATENCY=1 selects the geometries that are fastest for ONE step alone on the chip (q-split attention, the stage-3 GEMM
                # chain, the one-workgroup-per-CU C = 384 tail); the library reads it once, so the leg is a child run of this file
                import subprocess
                env = dict(os.environ, KVQ_LATENCY="1")
                cmd = [sys.executable, os.path.abspath(__file__), "--legs", "c2", "--streams", "1", "--no-cpu-baseline", "--no-pmc", "--profile-steps", "0",
                       "--steps", str(max(4, args.steps // 2)), "--warmup", "3", "--min-timed-s", "0.5", "--dtype", args.dtype, "--batch", str(B)] + _weights_arg()
                try:
                    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
                    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                    out["latency"] = {"value": d["value"], "unit": "videos/s", "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                                      "note": "KVQ_LATENCY=1 on one stream (child run): the launch geometries for one video at a time"}
                except Exception as e:  # noqa: BLE001
                    out["latency"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        for name, fn in (("c3", lambda pm: leg_c3(args, device, net, src, kd, pm)), ("c5", lambda pm: leg_c5(args, device, kd, pm)),
                         ("ksvqe", lambda pm: leg_ksvqe(args, device, kd, pm, 32)), ("ksvqe96", lambda pm: leg_ksvqe(args, device, kd, pm, 96))):
            if name in legs or (name == "ksvqe96" and "ksvqe" in legs):
                try:
                    pm = pmc_traffic(name, 3 if name.startswith("ksvqe") else 1, args.dtype, B) if want_pmc else None
                    out[name] = fn(pm)
                except Exception as e:  # noqa: BLE001  (an extra leg must not take the headline line down with it)
                    out[name] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
                torch.cuda.empty_cache()
        if "in_mix" in legs and want_pmc and out.get("roofline") is not None and headline_graph:
            out["roofline"]["in_mix"] = in_mix_evidence(args, B, nstream, out["ms_per_step"])
        # numbers of the extra legs where a record that keeps `config` and drops unknown top-level keys still holds them
        brief = {}
        for k in ("fp16_stress", "bf16_stress", "one_stream", "latency", "no_sampler", "two_launch_sampler", "batch8", "c3", "c5", "ksvqe", "ksvqe96"):
            v = out.get(k)
            if isinstance(v, dict) and "value" in v:
                brief[k] = {"value": round(v["value"], 2), "ms_per_step": round(v["ms_per_step"], 4)}
                for kk in ("max_abs_dscore_vs_reference_golden", "parity_ok", "max_abs_dscore_vs_fp16"):
                    if kk in v:
                        brief[k][kk] = v[kk]
                ws = v.get("whole_step") or {}
                if "frac" in ws:
                    brief[k]["whole_step_frac"] = round(ws["frac"], 4)
        out["config"]["legs"] = brief
        out["parity_pins"] = ("Swin3D trunk / heads / sampler / ResNet-50: oracle bit-pinned to the imported reference; UNPINNED by "
                              "necessity (packages absent here): SlowFast-R50 (pytorchvideo), torchvision Resize, CONTRIQUE's "
                              "torchvision resnet50 (stand-in = the reference's own Bottleneck)")
        out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(cfg, wts, hw, args.cpu_clips)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
