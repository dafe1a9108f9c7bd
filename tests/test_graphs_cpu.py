"""coin_amd/graphs.py without a GPU: the engine-free backward traversal that a backward graph is captured with must compute what
torch.autograd.grad computes -- custom Functions with non-tensor / frozen / tuple arguments, unused outputs (materialised zeros), a
parameter used twice, broadcast operands (the engine's shape reduction), dtype casts, unused targets."""
import torch

from coin_amd.graphs import _backward_on_this_thread


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, frozen, w, flag=None, tup=None):
        ctx.save_for_backward(x, w)
        ctx.k = k
        return x * w * k + frozen, x + 1

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g1, g2):
        x, w = ctx.saved_tensors
        return g1 * w * ctx.k + g2, None, None, g1 * x * ctx.k, None, None


def test_engine_free_backward_equals_autograd_grad():
    torch.manual_seed(0)
    lin, bn = torch.nn.Linear(8, 8), torch.nn.BatchNorm1d(8)
    w, unused, frozen = torch.randn(8, requires_grad=True), torch.randn(3, requires_grad=True), torch.randn(5, 8)
    x = torch.randn(5, 8, requires_grad=True)
    a, b = _Fn.apply(lin(x), 2.0, frozen, w, "flag", (frozen, 3))
    a2, _unused_output = _Fn.apply(a, 0.5, frozen, w)
    c = bn(a2) + b.chunk(2, dim=1)[0].sum() + lin(x).relu().to(torch.float64).float()
    outs = (c, (a * 2).detach(), b.sum(dim=1), x * 1.0)
    req = [o for o in outs if o.requires_grad]
    gs = [torch.randn_like(o) for o in req]
    wrt = [x, w, unused] + list(lin.parameters()) + list(bn.parameters())
    ref = torch.autograd.grad(req, wrt, gs, allow_unused=True, retain_graph=True)
    got = _backward_on_this_thread(req, gs, wrt)
    assert got[2] is None and ref[2] is None
    for r, g in zip(ref, got):
        assert (r is None) == (g is None)
        if r is not None:
            assert r.shape == g.shape and r.dtype == g.dtype
            torch.testing.assert_close(g, r, rtol=1e-6, atol=1e-6)
    # a second traversal of the same graph (a capture keeps the graph alive) gives the same answer
    again = _backward_on_this_thread(req, gs, wrt)
    assert all((p is None and q is None) or torch.equal(p, q) for p, q in zip(got, again))


def test_engine_free_backward_through_the_tiny_detectors_backbone():
    """The real module code (Bottleneck stacks, shimmed kernels on the CPU): gradients of every trainable backbone parameter and of the input."""
    from cpu_shim import cpu_kernels
    from e2e_util import tiny_product_detector

    torch.manual_seed(1)
    with cpu_kernels():
        bb = tiny_product_detector().backbone.train()
        x = torch.randn(2, 3, 64, 96, requires_grad=True)
        y = bb(x)["res4"]
        params = [p for p in bb.parameters() if p.requires_grad]
        gy = torch.randn_like(y)
        ref = torch.autograd.grad([y], [x] + params, [gy], allow_unused=True, retain_graph=True)
        got = _backward_on_this_thread([y], [gy], [x] + params)
    assert sum(r is not None for r in ref) > 10
    for r, g in zip(ref, got):
        assert (r is None) == (g is None)
        if r is not None:
            torch.testing.assert_close(g, r, rtol=1e-5, atol=1e-6)


def test_a_post_accumulate_hook_fires_for_an_undefined_gradient_with_the_directly_assigned_grad_in_place():
    """What coin_amd.graphs._Replay.backward relies on in data-parallel mode: it assigns the backward graph's static buffer to p.grad and
    returns None for the parameter; the engine still runs the parameter's accumulator node, whose post-accumulate hooks (the reducer's
    arrival counter) then see the assigned gradient -- exactly once."""
    import torch

    seen = []
    p = torch.nn.Parameter(torch.ones(3))
    p.register_post_accumulate_grad_hook(lambda q: seen.append(None if q.grad is None else q.grad.clone()))

    class Direct(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.w = w
            return x * 2

        @staticmethod
        def backward(ctx, g):
            ctx.w.grad = torch.full((3,), 5.0)
            return g * 2, None

    x = torch.ones(3, requires_grad=True)
    Direct.apply(x, p).sum().backward()
    assert len(seen) == 1 and torch.equal(seen[0], torch.full((3,), 5.0)) and torch.equal(p.grad, torch.full((3,), 5.0))
