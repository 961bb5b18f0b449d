#!/usr/bin/env python3
"""Where does the HOST spend an epoch of the rank rehearsal?  cProfile over tools/rehearse_rank.py:rehearse (expand="all").
    python tools/ab/profile_rehearsal_host.py [robot.xml]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import rehearse_rank as rr  # noqa: E402

robot = sys.argv[1] if len(sys.argv) > 1 else "xmls/ant.xml"
dev = torch.device("cuda", 0)
pr = cProfile.Profile()
pr.enable()
out = rr.rehearse(8, robot, 30, 6, "all", dev)
pr.disable()
print(out)
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
