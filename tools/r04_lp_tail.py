#!/usr/bin/env python3
"""Is the cfg5 label-propagation step (bench.py --workload labelprop) bound by the host's change-point search?  Runs the same bench
line with utils.change_point stubbed out (argument `nopelt`) or as shipped.
usage: python tools/r04_lp_tail.py [nopelt] -- <bench.py arguments>"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
args = sys.argv[1:]
nopelt = args and args[0] == "nopelt"
if nopelt:
    args = args[1:]
if args and args[0] == "--":
    args = args[1:]
sys.argv = ["bench.py"] + args
import bench  # noqa: E402  (puts the package on sys.path)
import utils as crw_utils  # noqa: E402
if nopelt:
    crw_utils.change_point = lambda *a, **k: None
bench.main()
