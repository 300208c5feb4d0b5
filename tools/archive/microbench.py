#!/usr/bin/env python3
"""Prints the roofline denominators measured on this GPU (stream copy/read, random line gathers)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from genedex_amd.device import measure_bandwidth  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
print(json.dumps(measure_bandwidth(torch.device("cuda", 0), gib=gib), indent=1))
