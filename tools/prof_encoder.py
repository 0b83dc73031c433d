"""VGG-16 encoder alone, a few passes (run under rocprofv3 --kernel-trace; tools/prof_calls.py lists the last pass call by call)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disn_amd import api, synth
from disn_amd.encoder import Vgg16Encoder
ctx = api.Context(0)
enc = Vgg16Encoder(ctx, synth.vgg_weights(0))
img = synth.image(seed=0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    g, f = enc(img)
torch.cuda.synchronize()
print("ok", float(g.sum()))
