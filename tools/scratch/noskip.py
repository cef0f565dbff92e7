#!/usr/bin/env python3
"""Bound of what fusing the decoder's skip copy into the UpConv launch could give: the step with the four
qpwc_copy_pixels launches of the concat removed (WRONG flows -- timing only).  QPWC_NOSKIP=1 removes them."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qpwcnet_amd import ops
if os.environ.get("QPWC_NOSKIP") == "1":
    ops.copy_pixels = lambda src, dst: dst
sys.argv = [sys.argv[0]] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "step_time.py"), run_name="__main__")
