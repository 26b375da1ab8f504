"""Test-only: lets TWO ranks share ONE MI355X (tests/test_gpu_dp_rccl.py::*_two_ranks_one_gpu_*).

RCCL refuses two ranks on one device, so the world-2 x real-GPU-sinks tests run their collectives over "gloo". Where this
torch build's gloo accepts device tensors nothing is patched. Where it does not, the `torch.distributed` entry points the
exchange objects call (`dp.LoRAGradArena`, `full_finetune.FullGradBuckets`, the workers' own checks) are wrapped so that
device tensors are staged through host copies: the D2H copy is stream-ordered behind the kernels that produced the
gradients (so "the collective saw the data the sinks wrote" is still what is being tested), the collective itself runs on
the host copy, the result is copied back. Product code is untouched and never imports this.
"""
import torch
import torch.distributed as dist


class _Done:
    def wait(self, *a, **k):
        return True

    def is_completed(self):
        return True


def gloo_takes_device_tensors(dev):
    t = torch.ones(8, device=dev)
    try:
        dist.all_reduce(t)
        torch.cuda.synchronize(dev)
        return float(t[0]) == float(dist.get_world_size())
    except Exception:
        return False


def install(dev):
    """Returns "native" (gloo handles device tensors) or "host-staged" (wrappers installed)."""
    if gloo_takes_device_tensors(dev):
        return "native"
    real_ar, real_ag_into, real_ag = dist.all_reduce, dist.all_gather_into_tensor, dist.all_gather

    def all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not tensor.is_cuda:
            return real_ar(tensor, op=op, group=group, async_op=async_op)
        host = tensor.detach().to("cpu")                  # synchronises with the producing stream
        real_ar(host, op=op, group=group)
        tensor.copy_(host)
        return _Done() if async_op else None

    def all_gather_into_tensor(output, input, group=None, async_op=False):
        if not output.is_cuda:
            return real_ag_into(output, input, group=group, async_op=async_op)
        hin = input.detach().to("cpu")
        hout = torch.empty(output.shape, dtype=output.dtype)
        real_ag_into(hout, hin, group=group)
        output.copy_(hout)
        return _Done() if async_op else None

    def all_gather(tensor_list, tensor, group=None, async_op=False):
        if not tensor.is_cuda:
            return real_ag(tensor_list, tensor, group=group, async_op=async_op)
        hl = [torch.empty(t.shape, dtype=t.dtype) for t in tensor_list]
        real_ag(hl, tensor.detach().to("cpu"), group=group)
        for d, h in zip(tensor_list, hl):
            d.copy_(h)
        return _Done() if async_op else None

    dist.all_reduce, dist.all_gather_into_tensor, dist.all_gather = all_reduce, all_gather_into_tensor, all_gather
    return "host-staged"
