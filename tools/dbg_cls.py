import sys; sys.path.insert(0, '.')
import torch
from egopack_amd import ops
from egopack_amd.models.tasks import RecognitionTask
ops.set_compute("bf16")
t = RecognitionTask(1024, 1024, (115, 478)).cuda()
x = torch.randn(2048, 1024, device="cuda").to(torch.bfloat16).requires_grad_(True)
y = torch.randint(0, 100, (2048, 2), device="cuda")
orig = ops._Linear.backward
def spy(ctx, dy):
    print("Linear.backward dy", dy.dtype, tuple(dy.shape), dy.stride(), dy.data_ptr() % 16, "x", ctx.saved_tensors[0].dtype)
    return orig(ctx, dy)
ops._Linear.backward = staticmethod(spy)
f = t.forward_features(x)
logits = t.forward_logits(f)
print("logits", [(l.dtype, l.stride()) for l in logits])
loss = ops.cross_entropy(tuple(logits), y)
ops.prof_reset(); ops.prof_enable(True)
loss.sum().backward()
torch.cuda.synchronize(); ops.prof_enable(False)
print({k: v["launches"] for k, v in ops.prof_report().items()})
