"""KSVQE (f1) forward timing on one MI355X: ms per forward at B samples of 32 frames, with a per-section breakdown
(CLIP_tool / QRS / CONTRIQUE / trunk stages / CDM) from HIP events.  `python tools/ksvqe_probe.py [B] [iters]`."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kvq_amd  # noqa
from kvq_amd.models import VQA_Network
from kvq_amd.models.backbones import ksvqe_modules as KM
from kvq_amd.utils import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
cfg = {"model": {"type": "KSVQE", "args": {"KSVQE": {"backbone": dict(checkpoint=True, pretrained=None, num_samples=1, sample_type="topkpertubation",
       CLIP_location=8, cls_use=True, tuning_stage=2, qls_swin=True, frozen3D=False, frozen_stages=-1),
       "head": {"in_channels": 768, "hidden_channels": 64}}}}}
net = VQA_Network(cfg)
sd = {"KSVQE_backbone." + k: torch.from_numpy(v) for k, v in synth.synth_ksvqe_weights(3).items()}
sd.update({"KSVQE_head." + k: torch.from_numpy(v) for k, v in synth.synth_vqa_head_weights(768, 64, 3, "stress").items()})
net.load_state_dict(sd, strict=False)
net = net.cuda().eval()
net.KSVQE_backbone.aux_loss = os.environ.get("KSVQE_AUX_LOSS", "0") == "1"      # the harness discards the loss (Trainer._score_all)
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_ksvqe_inputs(1, B, T).items()}

def run():
    with torch.no_grad():
        return net(inputs=dict(inp), reduce_scores=True)

for _ in range(3):
    s, _ = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(IT):
    s, _ = run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / IT
print(f"B={B} forward {dt*1e3:.2f} ms  -> {B/dt:.1f} samples/s ({T}-frame sample, 1 clip)", s.flatten().tolist()[:2])

# whole forwards of consecutive batches on alternating HIP streams (what Trainer._score_all does with KVQ_STREAMS=3)
NS = int(os.environ.get("KVQ_STREAMS", "3"))
if NS > 1 and os.environ.get("KSVQE_EAGER_STREAMS", "1") != "0":
    lanes = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(NS - 1)]
    def many(n):
        main = lanes[0]
        for st in lanes[1:]:
            st.wait_stream(main)
        outs = []
        for i in range(n):
            with torch.cuda.stream(lanes[i % NS]):
                outs.append(run()[0])
        for st in lanes[1:]:
            main.wait_stream(st)
        return outs
    many(2 * NS); torch.cuda.synchronize()
    t0 = time.perf_counter(); o = many(4 * IT); torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / (4 * IT)
    print(f"B={B} x {NS} streams: {dt2*1e3:.2f} ms per forward -> {B/dt2:.1f} samples/s; same scores: {torch.equal(o[0], s) and torch.equal(o[-1], s)}")

# hipGraph replay: capture one forward per lane (static inputs), replay lanes round-robin
if os.environ.get("KSVQE_GRAPH", "1") != "0":
    lanes = [torch.cuda.Stream() for _ in range(max(NS, 1))]
    graphs = []
    for st in lanes:
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            static = {k: v.clone() for k, v in inp.items()}
            for _ in range(2):
                with torch.no_grad():
                    net(inputs=dict(static), reduce_scores=True)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            with torch.no_grad():
                out = net(inputs=dict(static), reduce_scores=True)[0]
        graphs.append((g, static, out))
    torch.cuda.synchronize()
    def replay(n):
        for i in range(n):
            g, _, _ = graphs[i % len(lanes)]
            with torch.cuda.stream(lanes[i % len(lanes)]):
                g.replay()
    replay(2 * len(lanes)); torch.cuda.synchronize()
    t0 = time.perf_counter(); replay(len(lanes)); t_enq = (time.perf_counter() - t0) / len(lanes); torch.cuda.synchronize()
    print(f"host time of one graph launch: {t_enq*1e3:.2f} ms")
    t0 = time.perf_counter(); replay(4 * IT); torch.cuda.synchronize()
    dt3 = (time.perf_counter() - t0) / (4 * IT)
    print(f"B={B} hipGraph x {len(lanes)} lanes: {dt3*1e3:.2f} ms per forward -> {B/dt3:.1f} samples/s; same scores: {all(torch.equal(o, s) for _, _, o in graphs)}")

# section breakdown: wrap the submodules with event timers
bb = net.KSVQE_backbone
ev = []
def timed(name, fn):
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record(); ev.append((name, e0, e1)); return r
    return w
bb.CLIP_tool.forward = timed("clip", bb.CLIP_tool.forward)
bb.spa_patchnet.forward = timed("qrs", bb.spa_patchnet.forward)
bb.distortion_tool.forward = timed("contrique", bb.distortion_tool.forward)
bb.forward_stages = timed("trunk", bb.forward_stages)
bb._modulate = timed("cdm", bb._modulate)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
tot = {}
for n, a, b in ev:
    tot[n] = tot.get(n, 0) + a.elapsed_time(b)
print("sections ms:", {k: round(v, 3) for k, v in tot.items()}, "total", round(e0.elapsed_time(e1), 3))
