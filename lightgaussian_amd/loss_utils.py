"""Photometric loss of the training step on the HIP library.

Same names and argument meaning as the reference's utils/loss_utils.py:
  l1_loss(network_output, gt)                          <- utils/loss_utils.py:18-19
  ssim(img1, img2, window_size=11, size_average=True)  <- utils/loss_utils.py:46-85
plus the combination every trainer writes by hand (prune_finetune.py:161-164, distill_train.py:142-145,
train_densify_prune.py:135-138):
  l1_dssim_loss(image, gt, lambda_dssim) -> (loss, Ll1)

One lg_loss_forward launch produces BOTH means (and the three per-pixel partial-derivative maps the backward
filters); calling l1_loss(image, gt) and then ssim(image, gt) on the same tensors -- the reference's call
pattern -- runs the kernel once: the second call returns the other output of the same autograd node.
There is no PyTorch fallback: CPU tensors raise.
"""
import ctypes as C
import threading
import weakref

import torch

from . import _lib
from . import rasterizer as _rast


def _flags():
    f = 0
    if _rast.resolve_options().get("profile"):
        f |= _lib.FLAG_PROFILE
    return f


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, gt, l1_only=False):
        lib = _lib.load()
        ctx.l1_only = bool(l1_only)
        flags = _flags() | (_lib.FLAG_L1_ONLY if l1_only else 0)
        H, W = img.shape[-2], img.shape[-1]
        planes = img.numel() // (H * W)
        state = torch.empty(lib.lg_loss_state_bytes(planes, H, W), dtype=torch.uint8, device=img.device)
        out = torch.empty(2, dtype=torch.float32, device=img.device)
        stream = torch.cuda.current_stream(img.device).cuda_stream
        _lib.check(lib.lg_loss_forward(planes, H, W, img.data_ptr(), gt.data_ptr(), state.data_ptr(), out.data_ptr(), flags,
                                       C.c_void_p(stream)))
        ctx.save_for_backward(img, gt, state)
        ctx.set_materialize_grads(False)     # an unused output's gradient arrives as None (the backward handles it), not as a filled zero
        ctx.dims = (planes, H, W)
        ctx.token = {"consumed": False}      # shared with the memo of _both(): a graph that ran backward is not handed out again
        _memo.token = ctx.token              # forward runs on the caller's thread: handed to _both() through ITS thread-local slot
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        ctx.token["consumed"] = True
        img, gt, state = ctx.saved_tensors
        planes, H, W = ctx.dims
        lib = _lib.load()
        grad = torch.empty_like(img)
        keep = [None if g is None else g.contiguous().float() for g in (g_l1, g_ssim)]
        stream = torch.cuda.current_stream(img.device).cuda_stream
        _lib.check(lib.lg_loss_backward(planes, H, W, img.data_ptr(), gt.data_ptr(), state.data_ptr(),
                                        None if keep[0] is None else C.c_void_p(keep[0].data_ptr()), 1.0,
                                        None if keep[1] is None else C.c_void_p(keep[1].data_ptr()), 1.0,
                                        grad.data_ptr(), _flags() | (_lib.FLAG_L1_ONLY if ctx.l1_only else 0), C.c_void_p(stream)))
        return grad, None, None


def _prep(img, gt):
    if not (img.is_cuda and gt.is_cuda):
        raise RuntimeError("lightgaussian_amd.loss_utils runs on the MI355X HIP library only (no CPU fallback)")
    if img.shape != gt.shape or img.dim() not in (3, 4):
        raise ValueError(f"expected two [C,H,W] or [B,C,H,W] images of one shape, got {tuple(img.shape)} and {tuple(gt.shape)}")
    if gt.requires_grad:
        raise NotImplementedError("gradient with respect to the second image is not produced; detach() it (teacher / ground truth)")
    return img.contiguous().float(), gt.detach().contiguous().float()


_memo = threading.local()  # .last = (weakref(img), version, weakref(gt), version, l1, ssim, grad mode, token); per host thread


def _both(img, gt, last_use=False):
    """(l1, ssim) of one fused launch; memoised on tensor identity + version so that the reference's
    l1_loss(image, gt) ... ssim(image, gt) pair costs one forward and one backward launch.  The memo is dropped as soon as
    the pair is complete (last_use: ssim() and l1_dssim_loss() hand out the second output), so the image-sized saved state
    lives no longer than the caller's own references, and it is void once that graph has run backward: l1_loss(x, y)
    .backward() twice in a row recomputes, as the reference does."""
    last = getattr(_memo, "last", None)
    if last is not None:
        wi, vi, wg, vg, l1, ss, mode, token = last
        if (wi() is img and wg() is gt and vi == img._version and vg == gt._version and torch.is_grad_enabled() == mode
                and not token["consumed"]):
            if last_use:
                _memo.last = None
            return l1, ss
    a, b = _prep(img, gt)
    _memo.token = None
    l1, ss = _L1SSIM.apply(a, b)
    token = getattr(_memo, "token", None) if (torch.is_grad_enabled() and a.requires_grad) else None
    token = token if token is not None else {"consumed": False}
    _memo.last = None if last_use else (weakref.ref(img), img._version, weakref.ref(gt), gt._version, l1, ss,
                                        torch.is_grad_enabled(), token)
    return l1, ss


def l1_loss(network_output, gt):
    """utils/loss_utils.py:18-19: mean |network_output - gt|."""
    if _lazy_on():
        return _lazy_pair(network_output, gt, False, (0.0, 1.0, 0.0))
    return _both(network_output, gt)[0]


def l1_loss_only(network_output, gt):
    """mean |network_output - gt| for callers that never ask for SSIM: one streaming pass, no windowed moments
    (LG_FLAG_L1_ONLY).  l1_loss() is the right call inside the reference's trainers, which always follow it with ssim()."""
    a, b = _prep(network_output, gt)
    return _L1SSIM.apply(a, b, True)[0]


def l2_loss(network_output, gt):
    """utils/loss_utils.py:22-23 (not on the training path of the reference's trainers)."""
    return ((network_output - gt) ** 2).mean()


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:46-85: mean SSIM with the 11x11 Gaussian window (sigma 1.5), zero padding."""
    if window_size != 11 or not size_average:
        return _ssim_general(img1, img2, window_size, size_average)
    if _lazy_on():
        return _lazy_pair(img1, img2, True, (0.0, 0.0, 1.0))
    return _both(img1, img2, last_use=True)[1]


def _ssim_general(img1, img2, window_size, size_average):
    """The argument combinations no trainer of the reference uses (another window size, per-image means): the formula of
    utils/loss_utils.py:26-85 in plain torch ops on the images' device -- normalised Gaussian window of sigma 1.5, grouped
    conv2d with zero padding window_size // 2, C1 = 0.01^2, C2 = 0.03^2.  The fused HIP kernel covers the 11 x 11 / mean case."""
    import math
    import torch.nn.functional as F
    channel = img1.size(-3)
    g = torch.tensor([math.exp(-((x - window_size // 2) ** 2) / float(2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    window = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(channel, 1, window_size, window_size).contiguous().to(img1.device).type_as(img1)
    conv = lambda t: F.conv2d(t, window, padding=window_size // 2, groups=channel)  # noqa: E731
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq, sigma2_sq, sigma12 = conv(img1 * img1) - mu1_sq, conv(img2 * img2) - mu2_sq, conv(img1 * img2) - mu1_mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + c1) * (2 * sigma12 + c2)) / ((mu1_sq + mu2_sq + c1) * (sigma1_sq + sigma2_sq + c2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


def l1_dssim_loss(image, gt, lambda_dssim):
    """loss = (1 - lambda) * L1 + lambda * (1 - SSIM)  (prune_finetune.py:161-164).  Returns (loss, Ll1)."""
    l1, ss = _both(image, gt, last_use=True)
    return (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ss), l1


# ---- lazy scalars (opt-in: set_lazy(True), `python -m lightgaussian_amd.run --lazy-loss`) -------------------------------------
# Every trainer of the reference writes   loss = (1.0 - lambda) * Ll1 + lambda * (1.0 - ssim(image, gt));  loss.backward()   and then
# loss.item() for its running average (prune_finetune.py:161-175, distill_train.py:142-153, train_densify_prune.py:135-150).  On 0-dim
# device tensors that line is four tiny torch kernels, their autograd mirrors are five more and a fill (9-10 launches of ~5 us each
# between the fused loss kernel and the backward blend), and the .item() drains the whole iteration before the host may enqueue the
# optimizer step.  In lazy mode l1_loss() / ssim() return a LazyLoss: a 0-dim float32 tensor subclass that only REMEMBERS
# c0 + c1 * l1 + c2 * ssim.  Multiplying by / adding Python numbers and adding two LazyLoss of the same node move the coefficients
# on the host; backward() hands c1 and c2 to the fused node as its two upstream gradients (cached device scalars: the same float32
# numbers torch's own mul / rsub backward would have produced, so dL/dimage is bit-identical); item() / float() read the two means
# from a pinned copy that was requested on a side stream right behind the loss kernel -- it does not wait for the backward.
# Anything else (comparison, isnan, mean(), arithmetic with a tensor, printing ...) materialises the real tensor with torch ops and
# carries on: slower, never different.
_LAZY = threading.local()


def set_lazy(on=True):
    """Lazy loss scalars for THIS host thread (see above).  Returns the previous setting."""
    prev = getattr(_LAZY, "on", False)
    _LAZY.on = bool(on)
    return prev


def _lazy_on():
    return getattr(_LAZY, "on", False)


_CONST = {}          # (device, value) -> 0-dim float32 device tensor; the handful of coefficients a trainer uses
_SIDE = {}           # device -> (side stream, [free pinned [2] buffers])
_SIDE_LOCK = threading.Lock()


def _const(dev, value):
    key = (dev, float(value))
    t = _CONST.get(key)
    if t is None:
        if len(_CONST) >= 256:
            _CONST.clear()
        t = torch.full((), float(value), dtype=torch.float32, device=dev)
        _CONST[key] = t
    return t


class _HostPair:
    """{mean |x - y|, mean ssim} of one fused forward on the host: the copy is requested on a side stream right behind the loss
    kernel, get() waits for that copy only."""

    def __init__(self, out):
        dev = out.device
        with _SIDE_LOCK:
            side, free = _SIDE.setdefault(dev, (torch.cuda.Stream(device=dev), []))
            self.buf = free.pop() if free else torch.empty(2, dtype=torch.float32, pin_memory=True)
        self.dev, self.vals = dev, None
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self.buf.copy_(out, non_blocking=True)
            self.ev = torch.cuda.Event()
            self.ev.record(side)
        out.record_stream(side)

    def get(self):
        if self.vals is None:
            self.ev.synchronize()
            self.vals = (float(self.buf[0]), float(self.buf[1]))
            with _SIDE_LOCK:
                _SIDE[self.dev][1].append(self.buf)
            self.buf = None
        return self.vals

    def __del__(self):
        try:
            if self.buf is not None and self.ev.query():
                with _SIDE_LOCK:
                    _SIDE[self.dev][1].append(self.buf)
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


_META_PROPS = {"shape", "dtype", "device", "requires_grad", "is_cuda", "ndim", "is_leaf", "layout", "is_sparse", "is_quantized", "is_meta", "names"}
_META_FUNCS = {"dim", "size", "numel", "ndimension", "is_floating_point", "is_complex", "element_size", "get_device", "stride"}


class LazyLoss(torch.Tensor):
    """c0 + c1 * l1 + c2 * ssim of ONE fused loss node, not computed until somebody looks (see the section comment)."""

    @staticmethod
    def __new__(cls, l1, ss, coeff, host=None):
        c = tuple(float(x) for x in coeff)
        req = bool((c[1] != 0.0 and l1.requires_grad) or (c[2] != 0.0 and ss.requires_grad))
        r = torch.Tensor._make_wrapper_subclass(cls, (), dtype=torch.float32, device=l1.device, requires_grad=req)
        r._lz = (l1, ss, c, host)
        return r

    def materialize(self):
        """The real tensor, by torch ops on the two outputs of the fused node (keeps the autograd graph)."""
        l1, ss, (c0, c1, c2), _ = self._lz
        terms = [t for t in ((c1 * l1) if c1 != 0.0 else None, (c2 * ss) if c2 != 0.0 else None) if t is not None]
        out = terms[0] if terms else torch.zeros((), dtype=torch.float32, device=l1.device)
        for t in terms[1:]:
            out = out + t
        return out + c0 if c0 != 0.0 or not terms else out

    def _with(self, c):
        l1, ss, _, host = self._lz
        return LazyLoss(l1, ss, c, host)

    def host_value(self):
        l1, ss, (c0, c1, c2), host = self._lz
        if host is None:
            return float(self.materialize().detach())
        a, b = host.get()
        return c0 + c1 * a + c2 * b

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # safety net: an ATen call that reached the dispatcher with a LazyLoss argument (nothing above is known to do so)
        from torch.utils._pytree import tree_map
        real = lambda x: x.materialize() if isinstance(x, LazyLoss) else x  # noqa: E731
        return func(*tree_map(real, args), **tree_map(real, kwargs or {}))

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        num = lambda v: isinstance(v, (int, float)) and not isinstance(v, bool)  # noqa: E731
        a0 = args[0] if args else None
        a1 = args[1] if len(args) > 1 else None
        if isinstance(a0, LazyLoss):
            c0, c1, c2 = a0._lz[2]
            if name == "__get__":
                prop = getattr(getattr(func, "__self__", None), "__name__", "")
                if prop in _META_PROPS:
                    with torch._C.DisableTorchFunctionSubclass():
                        return func(*args, **kwargs)
                if prop == "grad_fn":
                    return None
            if name in _META_FUNCS:
                with torch._C.DisableTorchFunctionSubclass():
                    return func(*args, **kwargs)
            if not kwargs and len(args) == 2:
                if name in ("mul", "__mul__", "__rmul__", "multiply") and num(a1):
                    return a0._with((c0 * a1, c1 * a1, c2 * a1))
                if name in ("div", "__truediv__", "true_divide") and num(a1) and a1 != 0:
                    return a0._with((c0 / a1, c1 / a1, c2 / a1))
                if name in ("add", "__add__", "__radd__"):
                    if num(a1):
                        return a0._with((c0 + a1, c1, c2))
                    if isinstance(a1, LazyLoss) and a1._lz[0] is a0._lz[0] and a1._lz[1] is a0._lz[1]:
                        d0, d1, d2 = a1._lz[2]
                        return a0._with((c0 + d0, c1 + d1, c2 + d2))
                if name in ("sub", "__sub__"):
                    if num(a1):
                        return a0._with((c0 - a1, c1, c2))
                    if isinstance(a1, LazyLoss) and a1._lz[0] is a0._lz[0] and a1._lz[1] is a0._lz[1]:
                        d0, d1, d2 = a1._lz[2]
                        return a0._with((c0 - d0, c1 - d1, c2 - d2))
                if name == "__rsub__" and num(a1):                       # a1 - self
                    return a0._with((a1 - c0, -c1, -c2))
            if len(args) == 1 and not kwargs:
                if name in ("neg", "__neg__", "negative"):
                    return a0._with((-c0, -c1, -c2))
                if name in ("item", "__float__", "tolist"):
                    return a0.host_value()
                if name == "detach":
                    l1, ss, c, host = a0._lz
                    return LazyLoss(l1.detach(), ss.detach(), c, host)
                if name in ("mean", "sum", "clone", "contiguous", "squeeze", "float"):
                    return a0._with((c0, c1, c2))
            if name == "backward" and len(args) == 1 and kwargs.get("gradient") is None:
                l1, ss, _, _ = a0._lz
                roots = [(t, c) for t, c in ((l1, c1), (ss, c2)) if c != 0.0 and t.requires_grad]
                if not roots:
                    raise RuntimeError("element 0 of tensors does not require grad and does not have a grad_fn")
                return torch.autograd.backward([t for t, _ in roots], [_const(t.device, c) for t, c in roots],
                                               retain_graph=kwargs.get("retain_graph"), create_graph=bool(kwargs.get("create_graph", False)),
                                               inputs=kwargs.get("inputs"))
        # everything else: the real tensors
        from torch.utils._pytree import tree_map
        real = lambda x: x.materialize() if isinstance(x, LazyLoss) else x  # noqa: E731
        with torch._C.DisableTorchFunctionSubclass():
            return func(*tree_map(real, args), **tree_map(real, kwargs))


def _lazy_pair(img, gt, last_use, coeff):
    """The LazyLoss for one output of the fused node of (img, gt); the two calls of the reference's pair share the node (and its
    one pinned copy) through the same memo as the eager path."""
    last = getattr(_memo, "last", None)
    fresh = not (last is not None and last[0]() is img and last[2]() is gt and last[1] == img._version and last[3] == gt._version
                 and torch.is_grad_enabled() == last[6] and not last[7]["consumed"])
    l1, ss = _both(img, gt, last_use=last_use)
    host = None
    if l1.is_cuda:
        if fresh or getattr(_memo, "host", None) is None or _memo.host[0] is not l1:
            base = l1._base if l1._base is not None else None             # l1 = out[0]: a view of the [2] output of the forward
            _memo.host = (l1, _HostPair(base if base is not None and base.numel() == 2 else torch.stack((l1.detach(), ss.detach()))))
        host = _memo.host[1]
        if last_use:
            _memo.host = None
    return LazyLoss(l1, ss, coeff, host)
