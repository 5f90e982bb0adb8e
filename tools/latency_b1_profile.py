#!/usr/bin/env python3
"""100 single-utterance eval forwards (f16, split-K small-launch path) for a rocprofv3 --kernel-trace --stats run:
which launches the 228 us are made of.   rocprofv3 --kernel-trace --stats -d out -- python tools/latency_b1_profile.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepspeaker_pytorch_amd.model import DeepSpeakerModel
from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict

dev = torch.device("cuda", 0)
sd = synthetic_state_dict(0, 8)
low = "--eager-tiles" not in sys.argv
model = DeepSpeakerModel(512, 8, precision="f16", low_latency=low)
model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
model = model.to(dev).eval()
x = torch.randn(1, 1, 160, 64, device=dev)
with torch.no_grad():
    for _ in range(120):
        model(x)
torch.cuda.synchronize()
