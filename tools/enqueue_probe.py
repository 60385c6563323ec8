import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from kvq_amd.utils import synth
dev = torch.device("cuda:0")
net, cfg, wts, hw = bench.build_net("fp16", dev)
x = torch.from_numpy(synth.synth_clip(1234, 32, 224, 224, batch=4)).to(dev)
inp = {"technical": x}
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(2)]
with torch.no_grad():
    for st in streams:
        with torch.cuda.stream(st):
            net.swin_tiny_grpb_backbone.prepare(4, 32, 224, 224, dev)
            net(inputs=inp, reduce_scores=True)
    torch.cuda.synchronize()
    for n in (1, 3):
        t0 = time.perf_counter()
        for s in range(60):
            with torch.cuda.stream(streams[s % n]):
                net(inputs=inp, reduce_scores=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"streams {n}: enqueue {1e3*(t1-t0)/60:.3f} ms/step, total {1e3*(t2-t0)/60:.3f} ms/step")
