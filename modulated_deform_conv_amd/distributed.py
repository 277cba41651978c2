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


class FusedGradAllReduce:
    """Sum grad_weight and grad_bias over the data-parallel group with ONE collective.

    The flat fp32 buffer is allocated once and reused; gradients of other dtypes (fp16) are
    reduced in fp32 and cast back, so the result matches a single-GPU fp32 accumulation."""

    def __init__(self, group=None):
        self.group = group
        self._flat = None
        self._comm = None
        self._grouped = None   # None = untried, True / False = the grouped in-place launch works / does not

    def _buffer(self, numel, device, dtype):
        dtype = torch.float64 if dtype == torch.float64 else torch.float32   # never below fp32
        f = self._flat
        if f is None or f.numel() != numel or f.device != device or f.dtype != dtype:
            self._flat = torch.empty(numel, dtype=dtype, device=device)
        return self._flat

    def __call__(self, grad_weight, grad_bias=None, async_op=False):
        grads = [g for g in (grad_weight, grad_bias) if g is not None and g.numel() > 0]
        # RCCL, fp32 / fp64 gradients: both tensors in ONE grouped launch (ncclGroupStart / End), in place --
        # no flat staging buffer, i.e. four copy kernels and their launches less per step (the exchange is
        # 2.4 MB at cfg2: launch latency is all it costs)
        if (not async_op and grads and self._grouped is not False
                and all(g.is_cuda and g.is_contiguous() and g.dtype in (torch.float32, torch.float64) for g in grads)
                and hasattr(dist, "_coalescing_manager") and dist.get_backend(self.group) == "nccl"):
            # `_coalescing_manager` is a private torch API whose signature has changed between releases (2.0:
            # (group, device, reqs)); a TypeError / AttributeError is raised on ENTERING the context, before any
            # collective is issued, so falling back to the flat buffer below is safe -- and remembered
            try:
                cm = dist._coalescing_manager(group=self.group, device=grads[0].device, async_ops=False)
            except (TypeError, AttributeError):
                cm, self._grouped = None, False
            if cm is not None:
                with cm:
                    for g in grads:
                        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
                self._grouped = True
                return None
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
        for g in (grad_weight, grad_bias):
            if g is not None:
                g.record_stream(self._comm)
        main.wait_stream(self._comm)
        return None


def allreduce_module_grads(module, group=None):
    """Convenience for nn.Modules of this package after ``loss.backward()``."""
    FusedGradAllReduce(group)(module.weight.grad, module.bias.grad if module.bias is not None else None)
