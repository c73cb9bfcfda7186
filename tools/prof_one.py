#!/usr/bin/env python3
"""Run one prefill kernel shape a few times (for rocprofv3 counter passes).
usage: prof_one.py swa|gla [seq_len]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import tune_prefill as tp  # noqa: E402

if __name__ == "__main__":
    which = sys.argv[1]
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    if which == "swa":
        tp.swa([L], 28, 4, 0, 4096)
    else:
        tp.gla([L], 28, 4, 0)
