"""End-to-end harness throughput for the SimpleVQA config on a fake tree (uint8 .npy frame stacks + SlowFast feature files):
eager streams vs hipGraph replay lanes.  `python tools/harness_probe_simple.py [N] [T] [H] [W]`."""
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
for i in range(N):
    np.save(os.path.join(tmp, f"clip{i}.mp4.npy"), np.roll(base, i, axis=1))
    os.makedirs(os.path.join(tmp, "feat", f"clip{i}.mp4"))
    for k in range(8):
        np.save(os.path.join(tmp, "feat", f"clip{i}.mp4", f"feature_{k}_slow_feature.npy"), g.standard_normal((1, 2048, 1, 1, 1)).astype(np.float32))
        np.save(os.path.join(tmp, "feat", f"clip{i}.mp4", f"feature_{k}_fast_feature.npy"), g.standard_normal((1, 256, 1, 1, 1)).astype(np.float32))
open(os.path.join(tmp, "anno.csv"), "w").write("filename,score\n" + "".join(f"clip{i}.mp4,3.0\n" for i in range(N)))
cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "kwai_simpleVQA_test.yml")))
cfg["data"]["val"]["args"].update(anno_file=os.path.join(tmp, "anno.csv"), data_prefix=tmp, data_prefix_3D=os.path.join(tmp, "feat"))
cfg["load_path"] = cfg["test_load_path"] = None
os.chdir(tmp)
tr = Trainer(argparse.Namespace(opt="-", target_set="val", gpu_id="0"), cfg)
sd = {"simpleVQA_backbone." + k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_resnet50_weights(4, "stress").items()}
sd.update({"simpleVQA_head." + k: torch.from_numpy(v) for k, v in synth.synth_simple_head_weights(9472, 128, 4, "stress").items()})
tr.model.load_state_dict(sd, strict=False)
print(f"{N} videos of {T}x{H}x{W}, SimpleVQA (8 frames of 448x448 per video + 8x2304 SlowFast features from disk)")
for label, env in (("eager 3 streams + prefetch", dict(KVQ_GRAPH="0", KVQ_PREFETCH="2", KVQ_STREAMS="3")), ("eager, in-line", dict(KVQ_GRAPH="0", KVQ_PREFETCH="0", KVQ_STREAMS="3")),
                   ("graph 4 lanes, in-line", dict(KVQ_GRAPH="1", KVQ_PREFETCH="0", KVQ_STREAMS="4")), ("graph 4 lanes + prefetch", dict(KVQ_GRAPH="1", KVQ_PREFETCH="2", KVQ_STREAMS="4"))):
    os.environ.update(env)
    tr._score_all(); torch.cuda.synchronize()
    t0 = time.perf_counter(); s = tr._score_all(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label:28s} {N/dt:7.2f} videos/s  ({1e3*dt/N:.1f} ms per video)  checksum {float(np.sum(s)):.4f}")
