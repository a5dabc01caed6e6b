"""One launch of the motion kernel (B=4096, open box, float64, Philox) for counter collection."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab
np.random.seed(0)
walls = [] if len(sys.argv) < 2 or sys.argv[1] != "maze" else [[[.2, 0], [.2, .4]], [[.4, 1], [.4, .6]], [[.6, 0], [.6, .4]], [[.8, 1], [.8, .6]], [[.3, .5], [.7, .5]]]
ag = riab.Agent(riab.Environment({"walls": walls}), {"n_agents": 4096, "dt": 0.01})
hist = torch.empty((256, 8, 4096), dtype=torch.float32, device="cuda")
for _ in range(3):
    ag._advance(256, None, None, 1, {}, hist_view=hist)
torch.cuda.synchronize()
