#!/usr/bin/env python3
"""Runs tests/test_live_reference_sweep.py with its per-case lines printed (GPU box).  Keep the output under profiles/:
    python tools/twin_sweep.py > gpurun_out/twin_sweep.log"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.exit(subprocess.call([sys.executable, "-u", "-m", "pytest", os.path.join(ROOT, "tests", "test_live_reference_sweep.py"),
                          "-m", "gpu", "-q", "-s", "--timeout", "300", "--timeout-method", "thread"], cwd=ROOT))
