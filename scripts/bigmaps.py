import faulthandler; faulthandler.enable()
import sys, torch; sys.path.insert(0, ".")
from neighbour_feature_pooling_amd import NFPPooling, _abi
from neighbour_feature_pooling_amd._host import nfp_host
from bench import time_kernel_graph
L = _abi.load(); s = torch.cuda.Stream()
for shape in [(64,64,56,56),(64,128,28,28),(64,256,14,14),(8,40,28,28),(8,24,56,56),(4,16,112,112),(8,112,14,14),(2,8,100,140)]:
    m = NFPPooling(shape[1], R=1, measure="cosine", padding=1)
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    out = m(x); v = L.nfp_last_variant().decode()
    go = torch.randn_like(out)
    gx, = torch.autograd.grad(out, x, go); v += "/" + L.nfp_last_variant().decode()
    x2 = x.detach().double().requires_grad_(True)
    ref = nfp_host(x2, m.config); gref, = torch.autograd.grad(ref, x2, go.double())
    eo = (out.double()-ref).abs().max().item()/ref.abs().max().item(); eg = (gx.double()-gref).abs().max().item()/gref.abs().max().item()
    def ev(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
    out = m(x)
    tf = ev(lambda: m(x)); tb = ev(lambda: torch.autograd.grad(out, x, go, retain_graph=True))
    print(f"{shape}: {v:28s} err out {eo:.1e} grad {eg:.1e}  fwd {tf:8.1f} us bwd {tb:8.1f} us (eager, host-inclusive)")
