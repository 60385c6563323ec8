"""Host time to enqueue one C2 step (sampler inside the embedding launch) on each of the four lanes right after a synchronisation —
what sets the ramp of a timed block (lane k starts k steps' enqueue time after lane 0).  gpurun: python tools/enqueue_probe2.py"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
net, cfg, wts, hw = bench.build_net("fp16", dev)
B = 4
src = bench.Source(32, dev, 1234)
lanes = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(3)]
with torch.no_grad():
    for st in lanes:
        with torch.cuda.stream(st):
            net.swin_tiny_grpb_backbone.prepare(B, 32, 224, 224, dev)
    for s in range(8):
        with torch.cuda.stream(lanes[s % 4]):
            net(inputs={"technical": src.fragments(s * B, B)}, reduce_scores=True)
    torch.cuda.synchronize()
    for trial in range(3):
        ts = []
        t0 = time.perf_counter()
        for s in range(12):
            with torch.cuda.stream(lanes[s % 4]):
                a = time.perf_counter()
                f = src.fragments(s * B, B)
                b = time.perf_counter()
                net(inputs={"technical": f}, reduce_scores=True)
                ts.append((b - a, time.perf_counter() - b))
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("trial", trial, "per-step host us (fragments(), forward):", " ".join(f"{1e6*x:.0f}+{1e6*y:.0f}" for x, y in ts))
        print(f"   12 steps enqueued in {1e3*(t1-t0):.2f} ms, finished in {1e3*(t2-t0):.2f} ms")
    # the C call alone (plan, weights and bias image ready): time kvq_swin3d_forward_fragments through the module with the Python around it
    import cProfile, pstats, io
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for s in range(8):
        with torch.cuda.stream(lanes[s % 4]):
            net(inputs={"technical": src.fragments(s * B, B)}, reduce_scores=True)
    pr.disable()
    torch.cuda.synchronize()
    so = io.StringIO()
    pstats.Stats(pr, stream=so).sort_stats("cumulative").print_stats(18)
    print(so.getvalue()[:4000])
