#!/usr/bin/env python3
"""Entry point mirroring the reference's PointNetGPD/main_1v.py (same CLI); see mains.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnetgpd_amd.mains import run  # noqa: E402

if __name__ == "__main__":
    run("1v")
