"""One-off check (run in a CHILD process: a failure of the detection would be a segfault): a step function that stashes
its outputs -- autograd history and all -- in a dict between calls must make GraphedStep raise before it captures."""
import os
import subprocess
import sys

CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from collab_splats_amd import graphs, rasterization, MisplatError
from collab_splats_amd.synthetic import random_scene
dev = torch.device("cuda", 0)
W, H = 160, 96
sc = random_scene(3000, W, H, seed=2)
leaves = [sc[k].to(dev).requires_grad_(True) for k in ("means", "quats", "log_scales", "opacity_logits", "sh")]
V, K = sc["viewmats"].to(dev), sc["Ks"].to(dev)
stash = {}
def step():
    for l in leaves: l.grad = None
    out = rasterization(leaves[0], leaves[1], torch.exp(leaves[2]), torch.sigmoid(leaves[3]), leaves[4], V, K, W, H, sh_degree=3,
                        render_mode="RGB+ED", return_depth_normal=True)
    sum(o.sum() for o in out[:5]).backward()
    stash["meta"] = out[5]                     # keeps the graph of this call alive
step(); torch.cuda.synchronize()               # an eager call first: AccumulateGrad nodes on the default stream stay alive
try:
    graphs.GraphedStep(step, capacity=400000)
    print("NOT DETECTED (capture went through)")
except MisplatError as e:
    print("DETECTED:", str(e)[:90])
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, "-c", CHILD % root], capture_output=True, text=True, timeout=300)
print("child rc", r.returncode)
print(r.stdout[-400:])
if r.returncode != 0:
    print(r.stderr[-600:])
