#!/usr/bin/env python3
"""Replay of a captured forward with a camera updated in place: which output goes stale first?"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgaussian_amd import synthetic as syn, rasterizer
from lightgaussian_amd.gaussian_renderer import render

thrash = len(sys.argv) > 1 and sys.argv[1] == "thrash"
dev = torch.device("cuda:0")
g = syn.make_gaussians(12000, seed=5, log_scale_mean=math.log(0.04)).to(dev)
cam = syn.orbit_camera(1, 6, 256, 160).to(dev)
seq = [3, 5, 5, 0, 0, 1]
other = {k: syn.orbit_camera(k, 6, 256, 160).to(dev) for k in set(seq)}
pipe, bg = syn.PipelineParams(), torch.zeros(3, device=dev)
names = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")
pc = syn.SyntheticGaussians(*[getattr(g, n).detach().clone() for n in names], 3, 3)


def setcam(o):
    cam.world_view_transform.copy_(o.world_view_transform); cam.full_proj_transform.copy_(o.full_proj_transform); cam.camera_center.copy_(o.camera_center)


def body():
    with torch.no_grad():
        return render(cam, pc, pipe, bg)


keep = syn.orbit_camera(1, 6, 256, 160).to(dev)
body()
rasterizer.set_option("sync_free", True)
refs = {}
for k, o in other.items():
    setcam(o)
    out = body()
    st = rasterizer._PENDING.items[-1][0].clone()
    refs[k] = (out["render"].clone(), out["radii"].clone(), st.tolist())
rasterizer.pending_status()
setcam(keep)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body(); body()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
rasterizer.pending_status()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = body()
status = rasterizer._PENDING.items[-1][0]
big = torch.empty(1 << 28, dtype=torch.uint8, device=dev) if thrash else None
for k in seq:
    setcam(other[k])
    if thrash:
        big.zero_(); torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    r = refs[k]
    print("cam", k, "image equal", torch.equal(out["render"], r[0]), "radii equal", torch.equal(out["radii"], r[1]),
          "status", status.tolist(), "ref", r[2], flush=True)
