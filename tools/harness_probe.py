"""End-to-end harness throughput on a fake decoded-frame tree: N videos as uint8 [T,H,W,3] .npy stacks on
local disk -> ViewDecompositionDataset_KVQ (uint8 H2D, device-side samplers) -> KSVQE -> scores, with the input pipeline
in line (KVQ_PREFETCH=0) and prefetched by the host thread (default).  `python tools/harness_probe.py [N] [T] [H] [W]`."""
import os, sys, time, tempfile, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, yaml
import kvq_amd  # noqa
from kvq_amd.trainer import Trainer
from kvq_amd.utils import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
H = int(sys.argv[3]) if len(sys.argv) > 3 else 540
W = int(sys.argv[4]) if len(sys.argv) > 4 else 960
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="kvq_tree_")
g = np.random.Generator(np.random.PCG64(5))
base = g.integers(0, 256, size=(T, H, W, 3), dtype=np.uint8)
for i in range(N if not os.environ.get("HARNESS_SYNTH") else 1):
    np.save(os.path.join(tmp, f"clip{i}.mp4.npy"), np.roll(base, i, axis=1))
open(os.path.join(tmp, "anno.txt"), "w").write("".join(f"clip{i}.mp4,1,{i % 5},3.0\n" for i in range(N)))
cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "Kwai_KSVQE_test.yml")))
cfg["data"]["val"]["args"].update(anno_file=os.path.join(tmp, "anno.txt"), data_prefix=tmp)
os.chdir(tmp)
args = argparse.Namespace(opt="-", target_set="val", gpu_id="0")
tr = Trainer(args, cfg)
sd = {"KSVQE_backbone." + k: torch.from_numpy(v) for k, v in synth.synth_ksvqe_weights(3).items()}
sd.update({"KSVQE_head." + k: torch.from_numpy(v) for k, v in synth.synth_vqa_head_weights(768, 64, 3, "stress").items()})
tr.model.load_state_dict(sd, strict=False)
if os.environ.get("HARNESS_SYNTH"):
    # device-resident items (no decode / H2D): isolates the model side of Trainer._score_all (stream lanes, graph replay)
    class _Synth(torch.utils.data.Dataset):
        def __init__(self, n, t):
            self.items = [{k: (torch.from_numpy(v[0]).cuda() if v[0].ndim else int(v[0])) for k, v in synth.synth_ksvqe_inputs(s, 1, t).items()}
                          for s in range(4)]
            self.n = n
        def __len__(self):
            return self.n
        def __getitem__(self, i):
            return dict(self.items[i % 4])
    tr.val_dataset = _Synth(N, int(os.environ["HARNESS_SYNTH"]))
    for ns in (2, 3, 4, 5, 6):
        for pf in ("0", "2"):
            os.environ.update(KVQ_GRAPH="auto", KVQ_PREFETCH=pf, KVQ_STREAMS=str(ns))
            tr._score_all(); torch.cuda.synchronize()
            t0 = time.perf_counter(); s = tr._score_all(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"synthetic T={os.environ['HARNESS_SYNTH']} lanes={ns} prefetch={pf}: {N/dt:7.1f} videos/s ({1e3*dt/N:.2f} ms per video)")
    sys.exit(0)
print(f"{N} videos of {T}x{H}x{W} ({T*H*W*3/1e6:.0f} MB each), KSVQE, sample = 96 frames (3 x 32)")
for label, env in (("graph + prefetch 2", dict(KVQ_GRAPH="auto", KVQ_PREFETCH="2")), ("graph, in-line input", dict(KVQ_GRAPH="auto", KVQ_PREFETCH="0")),
                   ("eager + prefetch 2", dict(KVQ_GRAPH="0", KVQ_PREFETCH="2")), ("eager, in-line input", dict(KVQ_GRAPH="0", KVQ_PREFETCH="0"))):
    os.environ.update(env)
    tr._score_all(); torch.cuda.synchronize()                 # warm: plans, graphs are per call; page cache
    t0 = time.perf_counter(); s = tr._score_all(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label:24s} {N/dt:7.2f} videos/s  ({1e3*dt/N:.1f} ms per video)  checksum {float(np.sum(s)):.4f}")
