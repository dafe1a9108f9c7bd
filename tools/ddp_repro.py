#!/usr/bin/env python3
"""Development tool: is the 1-rank reducer test's plain run reproducible run to run?  (prints the per-step losses of plain / plain / reducer)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_ddp_gpu as T
runs = [T._run(False), T._run(False), T._run(True)]
for k in runs[0][0]:
    for step in range(3):
        print(step, k, *["%.7f" % r[step][k] for r in runs])
