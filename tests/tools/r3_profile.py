"""cProfile of Net.R3 (full 3C loop) on the small live net of tests/test_net_gpu.py."""
import cProfile
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for sub in ("channel-pruning_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import test_net_gpu as T  # noqa: E402
import lib.cfgs as cfgs  # noqa: E402

net, data = T.make_live_net(seed=2)
np.random.seed(11)
net.freeze_images(convs=net.convs)
cfgs.alpha = 1e-3
np.random.seed(99)
pr = cProfile.Profile()
pr.enable()
net.R3(rankdic={"conv1_2": 6, "conv2_1": 8, "conv2_2": 8, "conv3_1": 12})
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
