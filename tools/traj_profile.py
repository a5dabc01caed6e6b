"""Shader-clock profile of the four-wave trajectory kernel (a -DRIAB_T4_PROFILE build: tools/build_variants.sh prof):
per step, for workgroup 0: G wave: loop top -> walls done -> f taken -> next |v|^2 sent -> step handed over; S wave:
|v|^2 taken -> f sent.  RIAB_HIP_LIB=tools/exp/libt4_prof.so python tools/traj_profile.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ratinabox_amd as riab
L = riab._lib
np.random.seed(0)
ag = riab.Agent(riab.Environment({}), {"n_agents": 4096, "dt": 0.01})
T = 128
hist = torch.empty((T, 8, 4096), dtype=torch.float32, device="cuda")
zout = torch.zeros((T, 2, 4096), dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream(); raw = L.C.c_void_p(s.cuda_stream)
for _ in range(3):
    ag._advance(T, None, None, 1, {}, hist_view=hist, stream=raw, z_out=zout)
torch.cuda.synchronize()
p = zout.view(torch.int64).reshape(-1)[:T * 8].reshape(T, 8).cpu().numpy().astype(np.float64)
g0, g1, g2, g3, g4, s0, s1 = (p[:, k] for k in range(7))
sl = slice(40, 120)
def m(x): return "%7.0f" % np.median(x[sl])
print("cycles (median over steps 40..120; shader clock ticks of s_memtime):")
print(" G: top->walls done", m(g1 - g0), " wait for f", m(g2 - g1), " f->v2 sent", m(g3 - g2), " v2 sent->handed over", m(g4 - g3), " step (top->top)", m(np.diff(g0)))
print(" S: v2 taken->f sent", m(s1 - s0), " f sent->next v2 taken", m(s0[1:] - s1[:-1]))
print(" hand-overs: f sent(S)->f taken(G)", m(g2 - s1), "  v2 sent(G)->v2 taken(S, next step)", m(s0[1:] - g3[:-1]))
