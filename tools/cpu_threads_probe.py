import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, kvq_amd
from kvq_amd.utils import synth
from oracle import swin3d_oracle as O
cfg = synth.SWIN_T_GRPB
wts = synth.synth_swin_weights(cfg, 0, "stress"); hw = synth.synth_vqa_head_weights(768, 64, 0, "stress")
x = torch.from_numpy(synth.synth_clip(1, 32, 224, 224, batch=1))
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    with torch.no_grad():
        t = time.time(); O.vqa_head(O.swin3d_trunk(x, wts, cfg), hw); dt = time.time() - t
    print(nt, "threads:", round(dt, 2), "s/clip", flush=True)
