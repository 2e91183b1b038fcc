"""torch.compile keeps the drop-in promise (missing #7): the reference's NFP is plain ATen ops and compiles
as part of the model's one graph; here CUDA tensors go through the registered custom ops of _ops.py (a single graph node
with a fake implementation and an autograd formula), CPU tensors through _host.nfp_host, which Dynamo traces natively."""
import pytest
import torch
import torch.nn as nn

from neighbour_feature_pooling_amd import NFPPooling, nfp_pooling, _ops
from neighbour_feature_pooling_amd.functional import NfpConfig


class Net(nn.Module):
    def __init__(self, pooled=False, device=None):
        super().__init__()
        with torch.device(device or "cpu"):
            self.conv = nn.Conv2d(3, 16, 3, padding=1)
            self.nfp = NFPPooling(16, R=1, measure="cosine", padding=1)
            params = {"num_ftrs": {"m": 16}, "Model_name": "m", "Dataset": "d", "num_classes": {"d": 4}, "input_size": 9}
            self.head = nfp_pooling(Params=params) if pooled else None
            self.fc = nn.Linear(16 if pooled else 8, 4)

    def forward(self, x):
        f = self.conv(x)
        return self.fc(self.head(f) if self.head is not None else self.nfp(f).mean((2, 3)))


@pytest.mark.parametrize("pooled", [False, True])
def test_cpu_model_holding_nfp_compiles_as_one_graph(pooled):
    torch.manual_seed(0)
    net = Net(pooled)
    x = torch.randn(2, 3, 9, 9)
    ref = net(x)
    ref.square().sum().backward()
    g_ref = net.conv.weight.grad.clone()
    net.zero_grad()
    y = torch.compile(net, fullgraph=True, backend="aot_eager")(x)      # fullgraph: any graph break raises
    y.square().sum().backward()
    assert torch.allclose(y, ref, atol=1e-6) and torch.allclose(net.conv.weight.grad, g_ref, atol=1e-5)


@pytest.mark.parametrize("pooled", [False, True])
def test_cuda_model_traces_to_one_graph_with_the_custom_op(pooled):
    """No GPU needed: fake CUDA tensors.  torch.export (strict: through Dynamo) must produce ONE graph in which NFP is a
    single nfp_amd node, with output shapes from the fake implementation."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        net = Net(pooled, device="cuda")
        x = torch.empty(2, 3, 9, 9, device="cuda")
        ep = torch.export.export(net, (x,), strict=True)
    targets = [str(n.target) for n in ep.graph.nodes if n.op == "call_function"]
    want = "nfp_amd.nfp_pool.default" if pooled else "nfp_amd.nfp.default"
    assert sum(want in t for t in targets) == 1, targets
    out = [n for n in ep.graph.nodes if n.op == "output"][0]
    assert tuple(out.args[0][0].meta["val"].shape) == (2, 4)


def test_fake_implementations_state_the_library_shapes():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(3, 16, 10, 7, device="cuda", dtype=torch.bfloat16)
        for cfg, n in ((NfpConfig(R=2, measure="norm", p=2, padding=2), 24), (NfpConfig(R=1, measure="cosine", padding=0), 8),
                       (NfpConfig(R=2, measure="cosine", padding=2, inner_R=1), 32)):
            out, saved = torch.ops.nfp_amd.nfp(x, *_ops.cfg_args(cfg), True)
            ho = 10 if cfg.padding else 8
            assert tuple(out.shape) == (3, n, ho, ho - 3) and out.dtype == torch.bfloat16 and out.device.type == "cuda"
            assert saved.numel() == (3 * 70 if cfg.measure == "cosine" else 0) and saved.dtype == torch.float32
            gx = torch.ops.nfp_amd.nfp_backward(x, out, saved, out, *_ops.cfg_args(cfg))
            assert gx.shape == x.shape and gx.dtype == x.dtype
        gap, nfpm, omap, sv = torch.ops.nfp_amd.nfp_pool(x, *_ops.cfg_args(NfpConfig(R=1, measure="cosine", padding=1)), False, True)
        assert tuple(gap.shape) == (0, 16) and tuple(nfpm.shape) == (3, 8) and tuple(omap.shape) == (3, 8, 10, 7)
        gap, nfpm, omap, sv = torch.ops.nfp_amd.nfp_pool(x, *_ops.cfg_args(NfpConfig(R=1, measure="cosine", padding=1)), True, False)
        assert tuple(gap.shape) == (3, 16) and omap.numel() == 0
