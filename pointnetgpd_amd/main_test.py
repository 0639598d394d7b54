#!/usr/bin/env python3
"""Mirror of the reference's PointNetGPD/main_test.py: load a pickled model and score one in-gripper
cloud with a 10-vote majority (main_test.py:72-95).  Unlike the reference, importing this module has
no side effect; ``load_model`` does what main_test.py:34-56 does at import time."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnetgpd_amd import install_reference_aliases  # noqa: E402
from pointnetgpd_amd.scoring import test_network  # noqa: E402,F401  (re-exported: kinect2grasp.py:33,479)

MODEL_FILES = {"100": "../data/pointgpd_chann3_local.model", "50": "../data/pointgpd_50_points.model",
               "3class": "../data/pointnetgpd_3class.model"}     # main_test.py:34-41


def build_parser():
    p = argparse.ArgumentParser(description="pointnetGPD")
    p.add_argument("--cuda", action="store_true", default=False)
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--load-model", type=str, default="../data/pointnetgpd_3class.model")
    p.add_argument("--show_final_grasp", action="store_true", default=False)
    p.add_argument("--tray_grasp", action="store_true", default=False)
    p.add_argument("--using_mp", action="store_true", default=False)
    p.add_argument("--model_type", type=str)
    return p


def load_model(path, cuda=False, gpu=0):
    """torch.load of a whole pickled module (reference or ours), DataParallel unwrapped (main_test.py:42-56)."""
    install_reference_aliases()
    device = torch.device("cuda", gpu if gpu != -1 else 0) if cuda else torch.device("cpu")
    model = torch.load(path, map_location=device, weights_only=False)
    if isinstance(model, torch.nn.DataParallel):
        model = model.module
    return model.to(device).eval()


def vote(model, local_pc, num_point=500, repeat=10):
    """main_test.py:78-92: resample to num_point (without replacement iff enough points), B=1 forward,
    majority vote (scipy.stats.mode semantics: the smallest label among the most frequent)."""
    predict = []
    for _ in range(repeat):
        replace = not (len(local_pc) >= num_point)
        local_pc = local_pc[np.random.choice(len(local_pc), num_point, replace=replace)]
        predict.append(int(test_network(model, local_pc)[0]))
    values, counts = np.unique(predict, return_counts=True)
    return predict, int(values[np.argmax(counts)])


def main(argv=None):
    args = build_parser().parse_args(argv)
    args.cuda = args.cuda if torch.cuda.is_available else False      # sic, main_test.py:27
    if args.model_type in MODEL_FILES:
        args.load_model = MODEL_FILES[args.model_type]
    else:
        print("Using default model file")
    model = load_model(args.load_model, args.cuda, args.gpu)
    print("load model {}".format(args.load_model))
    torch.set_grad_enabled(False)
    local_pc = np.random.random([500, 3])   # test only (main_test.py:81)
    predict, result = vote(model, local_pc)
    print("voting: ", predict)
    print("Test result:", result)


if __name__ == "__main__":
    main()
