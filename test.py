#!/usr/bin/env python
"""Drop-in for the reference's ``test.py`` (:18-40): same flags (-o/--opt, -t/--target_set, --gpu_id),
same flow (yaml -> Trainer(args, opt).inferece()), output.txt 'video_name,score' + the metric line.

    python test.py -o config/kwai_swin_grpb_synthetic_test.yml --gpu_id 0
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 test.py -o <yml>
"""
import argparse
import os
import sys

import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kvq_amd  # noqa: E402,F401
from kvq_amd.trainer import Trainer  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("-o", "--opt", type=str, default="config/kwai_swin_grpb_synthetic_test.yml",
                        help="the option file")
    parser.add_argument("-t", "--target_set", type=str, default="val", help="target_set")
    parser.add_argument("--gpu_id", type=str, default="0")
    args = parser.parse_args()
    with open(args.opt, "r") as f:
        opt = yaml.safe_load(f)
    trainer = Trainer(args, opt)
    trainer.inferece()


if __name__ == "__main__":
    main()
