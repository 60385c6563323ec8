"""SimpleVQA spatial branch (ResNet-50 trunk) timing, stem as implicit GEMM over the 8-channel packed input vs im2col + GEMM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kvq_amd  # noqa
from kvq_amd.models.backbones import simpleVQA_model as S
from kvq_amd.utils import synth
net = S.ResNet()
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.synth_resnet50_weights(4, "stress").items()}, strict=False)
net = net.cuda().eval()
for shape in ((8, 3, 448, 448), (64, 3, 224, 224)):
    x = torch.randn(shape, device="cuda")
    for env in ("1", "0"):
        os.environ["KVQ_STEM_IMPLICIT"] = env
        with torch.no_grad():
            for _ in range(3): net.features(x)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): net.features(x)
            torch.cuda.synchronize()
        print(f"{shape}: stem implicit={env}: {(time.perf_counter() - t) * 100:.2f} ms per forward")
