"""Batch-sharded multi-GPU execution of the hot path (one process per GPU, RCCL over xGMI).

The reference has no multi-GPU code; the path shards naturally along the
batch: every image's output, grad_input, grad_offset and grad_mask depend on that image only
(mdeformable_conv.cu:54, 64-66, 228), and only grad_weight / grad_bias sum over the batch
(mdeformable_conv.cu:436-444).  So the forward needs no communication and the backward needs
exactly ONE exchange: an all-reduce(SUM -- not mean: the result must equal the single-GPU one) of
the fused fp32 buffer [grad_weight || grad_bias] (2.36 MB at cfg2: latency-bound; one fused
buffer = one RCCL launch).  Backend "nccl" is RCCL on ROCm; the CPU tests use "gloo".
"""
import torch
import torch.distributed as dist


def shard_bounds(batch, world_size, rank):
    """Contiguous batch slice [lo, hi) of `rank`; the first batch % world_size ranks get one more."""
    base, rem = divmod(batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(tensors, world_size=None, rank=None):
    """Slice every tensor of `tensors` (dict or sequence; None entries pass through) along dim 0."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank

    def cut(t):
        if t is None:
            return None
        lo, hi = shard_bounds(t.shape[0], world_size, rank)
        return t[lo:hi].contiguous()

    if isinstance(tensors, dict):
        return {k: cut(v) for k, v in tensors.items()}
    return type(tensors)(cut(v) for v in tensors)


def fused_grad_buffers(weight, bias=None):
    """-> (grad_weight, grad_bias): views of ONE flat buffer [grad_weight || grad_bias] in the dtype of `weight`, so that
    the data-parallel exchange is a single plain all-reduce of memory the backward wrote in place -- no staging copies, no
    grouped launch ("a fused fp32 buffer").  The bindings allocate their weight gradients this way
    (MDCONV_CUDA.modulated_deform_conv2d_backward_cuda, the autograd Functions, ops.py); `fused_view` finds the flat buffer
    again from the two views.  `grad_bias` is a 0-element view when `bias` is None or empty."""
    nb = 0 if bias is None else bias.numel()
    flat = torch.empty(weight.numel() + nb, dtype=weight.dtype, device=weight.device)
    return flat[:weight.numel()].view(weight.shape), flat[weight.numel():]


def fused_view(grad_weight, grad_bias=None):
    """The flat 1-D view over [grad_weight || grad_bias] when the two tensors are contiguous neighbours in one storage
    (`fused_grad_buffers`), else None.  grad_bias may be None / empty (then: grad_weight alone, flattened)."""
    if not grad_weight.is_contiguous():
        return None
    if grad_bias is None or grad_bias.numel() == 0:
        return grad_weight.view(-1)
    if (grad_bias.dtype != grad_weight.dtype or grad_bias.device != grad_weight.device or not grad_bias.is_contiguous()
            or grad_bias.untyped_storage().data_ptr() != grad_weight.untyped_storage().data_ptr()
            or grad_weight.data_ptr() + grad_weight.numel() * grad_weight.element_size() != grad_bias.data_ptr()):
        return None
    return torch.as_strided(grad_weight, (grad_weight.numel() + grad_bias.numel(),), (1,))


class FusedGradAllReduce:
    """Sum grad_weight and grad_bias over the data-parallel group with ONE collective.

    Gradients that live in one fused buffer (`fused_grad_buffers`: what the bindings of this package allocate) are reduced
    IN PLACE by a single plain `all_reduce` (fp32 / fp64); 16-bit gradients are reduced in fp32 through one staging buffer
    and cast back, so the result matches a single-GPU fp32 accumulation.  Tensors that are not neighbours are packed into
    the staging buffer first.  Only public torch.distributed calls are used."""

    def __init__(self, group=None):
        self.group = group
        self._flat = None
        self._comm = None
        self.last_mode = None   # "in-place" | "staged": how the last call ran (tests, bench.py)

    def _buffer(self, numel, device, dtype):
        dtype = torch.float64 if dtype == torch.float64 else torch.float32   # never below fp32
        f = self._flat
        if f is None or f.numel() != numel or f.device != device or f.dtype != dtype:
            self._flat = torch.empty(numel, dtype=dtype, device=device)
        return self._flat

    def __call__(self, grad_weight, grad_bias=None, async_op=False):
        fused = fused_view(grad_weight, grad_bias)
        if fused is not None and fused.dtype in (torch.float32, torch.float64):
            self.last_mode = "in-place"
            work = dist.all_reduce(fused, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            return (work, lambda: None) if async_op else None
        self.last_mode = "staged"
        grads = [fused] if fused is not None else [g for g in (grad_weight, grad_bias) if g is not None and g.numel() > 0]
        flat = self._buffer(sum(g.numel() for g in grads), grads[0].device, grads[0].dtype)
        off = 0
        for g in grads:
            flat[off:off + g.numel()].copy_(g.reshape(-1))
            off += g.numel()
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

        def finish():
            off = 0
            for g in grads:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()

        if async_op:
            return work, finish
        finish()
        return None

    def reduce_overlapped(self, grad_weight, grad_bias=None):
        """All-reduce under the tail of the backward that produced the gradients.

        The backward enqueues GEMM-2 / grad_bias BEFORE the grad_input gather and records an event
        in between (include/mdconv.h: mdconv_stream_wait_weight_ready).  The collective is issued
        on a communication stream that waits for that event only, so it runs while the gather is
        still executing; the caller's stream re-joins afterwards ("issue it
        on a side stream as soon as GEMM-2/bias finish").  Call it after the backward that produced
        the gradients has been issued on the CURRENT stream -- directly through MDCONV_CUDA or via
        ``loss.backward()`` (autograd runs the op on a worker thread but on the forward's stream;
        the event is keyed by (device, stream), not by thread).  CPU tensors (gloo tests) take the
        plain path."""
        if not grad_weight.is_cuda:
            return self(grad_weight, grad_bias)
        from . import _capi
        if self._comm is None:
            self._comm = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        _capi.stream_wait_weight_ready(self._comm, producer=main)
        with torch.cuda.stream(self._comm):
            self(grad_weight, grad_bias)
        if not torch.cuda.is_current_stream_capturing():   # (a captured step keeps its tensors alive itself)
            for g in (grad_weight, grad_bias):
                if g is not None:
                    g.record_stream(self._comm)
        main.wait_stream(self._comm)
        return None


def allreduce_module_grads(module, group=None):
    """Convenience for nn.Modules of this package after ``loss.backward()``."""
    FusedGradAllReduce(group)(module.weight.grad, module.bias.grad if module.bias is not None else None)
