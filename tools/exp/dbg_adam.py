import sys; sys.path.insert(0, ".")
import torch
from egopack_amd.optim import FlatAdam
g = torch.Generator().manual_seed(5)
shapes = [(33, 16), (16,)]
ps = [torch.randn(s, generator=g) for s in shapes]
grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(5)]
def run(opt, params, its):
    for it in its:
        for p, gr in zip(params, grads[it]):
            if p.grad is None: p.grad = gr.clone().to(p.device)
            else: p.grad.copy_(gr)
        opt.step()
a = [p.clone().cuda().requires_grad_(True) for p in ps]
oa = FlatAdam(a, lr=1e-2, weight_decay=1e-3); run(oa, a, range(3))
sd = oa.state_dict()
b = [p.detach().clone().requires_grad_(True) for p in a]
ob = FlatAdam(b, lr=1.0); ob.load_state_dict(sd)
print("pg", ob.param_groups[0]["lr"], ob.param_groups[0]["weight_decay"], ob.param_groups[0]["betas"], ob.step_count)
print("p equal before", [torch.equal(x.detach(), y.detach()) for x, y in zip(a, b)])
run(oa, a, [3]); run(ob, b, [3])
print("m equal", torch.equal(oa.flat_m, ob.flat_m), "v equal", torch.equal(oa.flat_v, ob.flat_v), oa.step_count, ob.step_count)
print("p equal", [torch.equal(x.detach(), y.detach()) for x, y in zip(a, b)], (a[0]-b[0]).abs().max().item())
print("g equal", torch.equal(oa.flat_g, ob.flat_g))
