#!/usr/bin/env python3
"""The inner loop of prune_finetune.py / train_densify_prune.py (reference lines 144-170) on MI355X, on a synthetic scene:

    render(viewpoint_cam, gaussians, pipe, background)            -> image, viewspace_points, radii
    loss = (1 - lambda) * l1_loss(image, gt) + lambda * (1 - ssim(image, gt))
    loss.backward(); optimizer.step(); optimizer.zero_grad()

with this repo's drop-in pieces: gaussian_renderer.render (HIP rasterizer, getters evaluated in-kernel) and
loss_utils.l1_loss / ssim (one fused HIP launch for both).  The optimizer is torch.optim.Adam as in the reference
(scene/gaussian_model.py:training_setup).

    python examples/finetune_step.py [--n-gaussians 3000000] [--iters 100]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightgaussian_amd import synthetic as syn  # noqa: E402
from lightgaussian_amd.gaussian_renderer import render  # noqa: E402
from lightgaussian_amd.loss_utils import l1_loss, ssim  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-gaussians", type=int, default=1_000_000)
    ap.add_argument("--views", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--lambda-dssim", type=float, default=0.2)     # arguments/__init__.py
    ap.add_argument("--fused-adam", action="store_true", help="what `python -m lightgaussian_amd.run --fused-adam` switches on")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    truth = syn.make_gaussians(args.n_gaussians).to(dev)
    cams = [syn.orbit_camera(k, args.views, args.width, args.height).to(dev) for k in range(args.views)]
    bg = torch.zeros(3, device=dev)
    pipe = syn.PipelineParams()
    with torch.no_grad():
        gts = [render(c, truth, pipe, bg)["render"].clone() for c in cams]
    # the model being fine-tuned: the same scene with perturbed colours and opacities
    gen = torch.Generator().manual_seed(1)
    g = syn.make_gaussians(args.n_gaussians)
    g._features_dc += 0.1 * torch.randn(g._features_dc.shape, generator=gen)
    g._opacity += 0.3 * torch.randn(g._opacity.shape, generator=gen)
    g = g.to(dev).requires_grad_(True)
    if args.fused_adam:
        from lightgaussian_amd import run as lg_run
        lg_run.fused_adam(True)
    opt = torch.optim.Adam([{"params": [g._xyz], "lr": 1.6e-6}, {"params": [g._features_dc], "lr": 2.5e-3},
                            {"params": [g._features_rest], "lr": 2.5e-3 / 20.0}, {"params": [g._opacity], "lr": 0.05},
                            {"params": [g._scaling], "lr": 0.005}, {"params": [g._rotation], "lr": 0.001}], lr=0.0, eps=1e-15)
    first = last = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.iters):
        k = it % args.views
        pkg = render(cams[k], g, pipe, bg)
        image = pkg["render"]
        Ll1 = l1_loss(image, gts[k])
        loss = (1.0 - args.lambda_dssim) * Ll1 + args.lambda_dssim * (1.0 - ssim(image, gts[k]))
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        if it < args.views:
            first = float(loss.detach()) if first is None else first + float(loss.detach())
        if it >= args.iters - args.views:
            last = float(loss.detach()) if last is None else last + float(loss.detach())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = min(args.views, args.iters)
    print(f"{args.iters} iterations, {args.n_gaussians} Gaussians, {args.width}x{args.height}: {dt / args.iters * 1e3:.2f} ms/iteration "
          f"(render fwd+bwd + L1/SSIM + Adam{' fused' if args.fused_adam else ''}); mean loss first {n} iterations {first / n:.5f} -> last {n} {last / n:.5f}")
    assert last < first, "the loss did not go down"


if __name__ == "__main__":
    main()
