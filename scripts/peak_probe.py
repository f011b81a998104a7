#!/usr/bin/env python3
"""Measured MFMA and HBM peaks of this GPU (erfnet_pytorch_amd/probe.py); prints one JSON object.

    python scripts/peak_probe.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import probe  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = probe.measured_peaks(dev)
    # sweep of residency for the record: 1, 2 and 4+ rounds of workgroups
    out["mfma_bf16_by_workgroups"] = {str(w): probe.mfma_peak(1, dev, workgroups=w)[0] for w in (256, 512, 1024, 2048, 4096)}
    print(json.dumps(out))
