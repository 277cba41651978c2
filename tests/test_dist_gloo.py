"""world_size-2 gloo tests (CPU) of the multi-GPU plan: contiguous batch shards, no forward
communication, one fused all-reduce(SUM) of [grad_weight || grad_bias].  The compute on each rank is
the oracle (allowed in tests); the collective code is the product's distributed.py."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from modulated_deform_conv_amd.distributed import (FusedGradAllReduce, fused_grad_buffers, fused_view, shard_batch,
                                                   shard_bounds)
from tests.cases import _c, make_inputs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds_cover_batch():
    for batch in (1, 5, 8, 32, 33):
        for ws in (1, 2, 3, 8):
            got = [shard_bounds(batch, ws, r) for r in range(ws)]
            assert got[0][0] == 0 and got[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            assert max(h - l for l, h in got) - min(h - l for l, h in got) <= 1


def _worker(rank, world, port, case, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    t = make_inputs(case, dtype=torch.float64)
    args = (case["stride"], case["padding"], case["dilation"], case["groups"], case["dgroups"], 64)
    sh = shard_batch({k: t[k] for k in ("input", "offset", "mask", "grad_output")}, world, rank)
    out = oracle.forward(case["op"], sh["input"], t["weight"], t["bias"], sh["offset"], sh["mask"], *args)
    g = oracle.backward(case["op"], sh["input"], t["weight"], t["bias"], sh["offset"], sh["mask"],
                        sh["grad_output"], *args)
    # the ONE exchange of the path; on CPU tensors the overlapped entry takes the plain route
    FusedGradAllReduce().reduce_overlapped(g["grad_weight"], g["grad_bias"])
    if rank == 0:
        outs = [torch.empty_like(out) for _ in range(world)]
    gathered = [None] * world
    dist.all_gather_object(gathered, {k: g[k] for k in ("grad_input", "grad_offset", "grad_mask")} | {"out": out})
    if rank == 0:
        ret["grad_weight"], ret["grad_bias"] = g["grad_weight"], g["grad_bias"]
        for k in ("out", "grad_input", "grad_offset", "grad_mask"):
            ret[k] = torch.cat([x[k] for x in gathered], 0)
    dist.destroy_process_group()


def test_batch_sharded_equals_single_process():
    case = _c("dist", oracle.MDCN2D, 5, 6, 4, (7, 6), 3, dgroups=2, groups=2, seed=31)
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, case, ret), nprocs=world, join=True)
        ret = dict(ret)
    t = make_inputs(case, dtype=torch.float64)
    args = (1, 1, 1, 2, 2, 64)
    want_out = oracle.forward(case["op"], t["input"], t["weight"], t["bias"], t["offset"], t["mask"], *args)
    want = oracle.backward(case["op"], t["input"], t["weight"], t["bias"], t["offset"], t["mask"],
                           t["grad_output"], *args)
    assert torch.allclose(ret["out"], want_out, rtol=1e-12, atol=1e-12)
    for k in ("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"):
        assert torch.allclose(ret[k], want[k], rtol=1e-11, atol=1e-11), k


def _worker_fp16(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gw = torch.full((3, 2, 3, 3), float(rank + 1), dtype=torch.float16)
    gb = torch.full((3,), 0.5 * (rank + 1), dtype=torch.float16)
    red = FusedGradAllReduce()
    work, finish = red(gw, gb, async_op=True)
    work.wait()
    finish()
    if rank == 0:
        ret["gw"], ret["gb"] = gw, gb
    dist.destroy_process_group()


def test_fused_allreduce_async_and_fp16():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_fp16, args=(world, port, ret), nprocs=world, join=True)
        ret = dict(ret)
    assert torch.equal(ret["gw"], torch.full((3, 2, 3, 3), 3.0, dtype=torch.float16))
    assert torch.equal(ret["gb"], torch.full((3,), 1.5, dtype=torch.float16))


def _worker_fused(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, b = torch.zeros(3, 2, 3, 3), torch.zeros(3)
    red = FusedGradAllReduce()
    modes = []
    # (1) the layout the bindings allocate: one flat buffer, two views -> ONE in-place all-reduce, no staging
    gw, gb = fused_grad_buffers(w, b)
    gw.fill_(rank + 1.0); gb.fill_(10.0 * (rank + 1))
    base_ptr = gw.data_ptr()
    red(gw, gb)
    modes.append((red.last_mode, red._flat is None, gw.data_ptr() == base_ptr))
    # (2) no bias
    gw2, gb2 = fused_grad_buffers(w, None)
    gw2.fill_(rank + 1.0)
    red.reduce_overlapped(gw2, gb2)
    modes.append((red.last_mode, gb2.numel()))
    # (3) two unrelated tensors: packed into the staging buffer
    gw3, gb3 = torch.full_like(w, rank + 1.0), torch.full_like(b, 0.5)
    red(gw3, gb3)
    modes.append((red.last_mode,))
    # (4) fp16 neighbours: reduced in fp32 through the staging buffer (one copy in, one out)
    gw4, gb4 = fused_grad_buffers(w.half(), b.half())
    gw4.fill_(rank + 1.0); gb4.fill_(0.25)
    red(gw4, gb4)
    modes.append((red.last_mode, red._flat.dtype))
    if rank == 0:
        ret["modes"] = modes
        ret["vals"] = [gw.clone(), gb.clone(), gw2.clone(), gw3.clone(), gb3.clone(), gw4.clone(), gb4.clone()]
    dist.destroy_process_group()


def test_fused_buffer_is_reduced_in_place_by_one_plain_collective():
    """the backward writes grad_weight || grad_bias into ONE buffer (fused_grad_buffers, what
    MDCONV_CUDA / the autograd Functions allocate) and the exchange is a single `all_reduce` of it -- no staging copies,
    no private torch.distributed API."""
    assert fused_view(*fused_grad_buffers(torch.zeros(2, 3), torch.zeros(2))).numel() == 8
    assert fused_view(torch.zeros(2, 3), torch.zeros(2)) is None
    import inspect
    from modulated_deform_conv_amd import distributed
    assert "_coalescing_manager" not in inspect.getsource(distributed)
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_fused, args=(world, port, ret), nprocs=world, join=True)
        ret = dict(ret)
    m = ret["modes"]
    assert m[0] == ("in-place", True, True) and m[1] == ("in-place", 0) and m[2] == ("staged",)
    assert m[3] == ("staged", torch.float32)
    gw, gb, gw2, gw3, gb3, gw4, gb4 = ret["vals"]
    assert torch.all(gw == 3) and torch.all(gb == 30) and torch.all(gw2 == 3)
    assert torch.all(gw3 == 3) and torch.all(gb3 == 1) and torch.all(gw4 == 3) and torch.all(gb4 == 0.5)
