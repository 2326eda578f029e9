"""bf16 compute copies + fp32 master weights, without torch.autocast.

Why not autocast: on this model it launches ~2 200 tiny kernels per step (one weight cast per Linear in forward, one
bf16->fp32 gradient cast and one accumulate per parameter in backward: 761 tensors) and keeps LayerNorm in fp32 with
casts on both sides -- ~20 ms of a 170 ms step on MI355X (profiles/r01).  Here instead

  * the module's parameters ARE bf16 tensors, each a view into one flat bf16 buffer per gradient bucket; gradients
    are produced by autograd directly into flat bf16 buckets (grit_amd.ddp.BucketedDataParallel), which is also what
    RCCL all-reduces -- half the bytes on xGMI, no staging copy;
  * the optimizer owns fp32 master parameters laid out bucket-by-bucket the same way, so "bf16 grads -> fp32 grads"
    and "fp32 masters -> bf16 compute weights" are ONE copy kernel per bucket per step;
  * numerically sensitive spots stay fp32 by construction in the modules (softmax / LayerNorm statistics inside the
    HIP kernels, MSDA sampling locations, vocabulary logits + log-softmax, the loss).

State dicts are exported from the masters (fp32) under the reference's key names.
"""
import torch
from torch import nn

from grit_amd.ddp import BucketedDataParallel


class Bf16Compute(nn.Module):

    def __init__(self, module, bucket_mb=64, process_group=None):
        super().__init__()
        import torch.distributed as dist
        names = {p: n for n, p in module.named_parameters()}
        if dist.is_initialized() and dist.get_world_size(process_group) > 1:
            for p in module.parameters():  # rank 0's fp32 initialisation is THE model: masters must start from it
                dist.broadcast(p.data, src=0, group=process_group)
        fp32 = {p: p.detach().clone().float() for p in module.parameters() if p.requires_grad}
        module.to(torch.bfloat16)  # parameters and floating buffers; integer buffers untouched
        self.ddp = BucketedDataParallel(module, bucket_mb=bucket_mb, process_group=process_group, repack_unused=False,
                                        broadcast_parameters=False)
        self.module = module
        self._masters, self._pairs = [], []
        for b in self.ddp.buckets:
            n = b.flat.numel()
            compute_flat = torch.empty(n, dtype=torch.bfloat16, device=b.flat.device)
            master_flat = torch.empty(n, dtype=torch.float32, device=b.flat.device)
            master_grad = torch.zeros(n, dtype=torch.float32, device=b.flat.device)
            off = 0
            for p in b.params:
                k = p.numel()
                m = nn.Parameter(master_flat[off:off + k].view_as(p))
                m.data.copy_(fp32[p])
                m.grad = master_grad[off:off + k].view_as(p)
                compute_flat[off:off + k].view_as(p).copy_(m.data)
                p.data = compute_flat[off:off + k].view_as(p)  # the module now computes on the flat bf16 copy
                self._masters.append((names[p], m))
                off += k
            self._pairs.append((b, compute_flat, master_flat, master_grad))

    # ------------------------------------------------------------------ what the engine calls
    def forward(self, *args, **kwargs):
        return self.ddp(*args, **kwargs)  # resets .grad to None first: backward assigns, one packed copy per bucket

    def named_master_parameters(self):
        return list(self._masters)

    def finish_gradient_sync(self):
        self.ddp.finish_gradient_sync()
        for b, _, _, master_grad in self._pairs:
            master_grad.copy_(b.flat)  # bf16 -> fp32, one kernel per bucket

    def after_optimizer_step(self):
        for b, compute_flat, master_flat, _ in self._pairs:
            compute_flat.copy_(master_flat)  # fp32 -> bf16

    def master_state_dict(self):
        """fp32 state dict under the reference's key names (masters for trainable tensors, upcast copies otherwise)."""
        sd = {k: (v.float() if v.is_floating_point() else v).clone() for k, v in self.module.state_dict().items()}
        for name, m in self._masters:
            sd[name] = m.detach().clone()
        return sd

    @property
    def unused_parameters(self):
        return self.ddp.unused_parameters

    def gradient_bytes(self):
        return self.ddp.gradient_bytes()
