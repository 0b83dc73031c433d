import sys, numpy as np, torch
sys.path.insert(0, '.')
from disn_amd.api import Context
from oracle import net
c = Context(0)
rng = np.random.default_rng(0)
pts = rng.uniform(-1.2, 1.2, (5000, 3)).astype(np.float32)
got = c.project_points(net.GT_TRANS_MAT, pts).cpu().numpy()
np.save('gpurun_out/proj_gpu.npy', got)
