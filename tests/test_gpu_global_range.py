"""The multi-process global-range mode (collective 3) on the FUSED stage schedule: two processes
share the one GPU of the test box (gloo between them; RCCL wants a GPU per rank), each runs half of every batch through
FusedHotPath with the stage call split at its QuantActs (CDN_X_DEFER_RANGE + phases + cdn_quantact_commit_range), and
together they must track EXACTLY the ranges -- and produce exactly the outputs -- of one process running the whole
batches on the plain fused schedule."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batches(n, steps):
    g = torch.Generator().manual_seed(5)
    return [(torch.randn(n, 1024, 8, 8, generator=g).abs_() * (1.0 + 0.4 * i)) for i in range(steps)]


def _ranges(net):
    from codenet_amd.portable_quantizer.quant_modules import QuantAct
    acts = [m for m in net.modules() if isinstance(m, QuantAct)]
    return [(a.x_min.detach().cpu().clone(), a.x_max.detach().cpu().clone()) for a in acts]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from codenet_amd import pipeline
    net = pipeline.build_hot_path(quantized=True).cuda()
    assert pipeline.set_global_range(net, True) > 0
    assert pipeline.FusedHotPath.supported(net.deconv_layers, (3, 1024, 8, 8))
    fused = pipeline.FusedHotPath(net.deconv_layers)
    lo, hi = pipeline.shard_range(6, rank, world)
    outs = [fused(x[lo:hi].cuda()).cpu().clone() for x in _batches(6, 3)]
    torch.cuda.synchronize()
    torch.save({"outs": outs, "ranges": _ranges(net), "shard": (lo, hi)}, out % rank)
    dist.destroy_process_group()


def test_fused_stages_in_global_range_mode_equal_one_process_world2(tmp_path):
    world = 2
    out = str(tmp_path / "g%d.pt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = [torch.load(out % r) for r in range(world)]
    from codenet_amd import pipeline
    net = pipeline.build_hot_path(quantized=True).cuda()
    fused = pipeline.FusedHotPath(net.deconv_layers)
    ref_outs = [fused(x.cuda()).cpu().clone() for x in _batches(6, 3)]
    ref_ranges = _ranges(net)
    for r in res:
        lo, hi = r["shard"]
        for (a_lo, a_hi), (b_lo, b_hi) in zip(r["ranges"], ref_ranges):
            assert torch.equal(a_lo, b_lo) and torch.equal(a_hi, b_hi)
        for a, b in zip(r["outs"], ref_outs):
            assert torch.equal(a, b[lo:hi])
    # the mode is what makes them equal: a rank alone on its half tracks other ranges
    own = pipeline.build_hot_path(quantized=True).cuda()
    f2 = pipeline.FusedHotPath(own.deconv_layers)
    for x in _batches(6, 3):
        f2(x[:3].cuda())
    assert any(not torch.equal(a[0], b[0]) or not torch.equal(a[1], b[1]) for a, b in zip(_ranges(own), ref_ranges))


def test_split_stage_call_equals_the_plain_call_on_one_rank():
    """The split call by itself (no process group needed): DEFER_RANGE + the three phases + cdn_quantact_commit_range with
    this rank's own extremes = the plain fused call, bit for bit, over three batches (initialisation + two momentum
    steps); and the argument checks of the new flags."""
    from codenet_amd import pipeline, _native as N_
    net_a = pipeline.build_hot_path(quantized=True).cuda()
    net_b = pipeline.build_hot_path(quantized=True).cuda()
    pipeline.set_global_range(net_b, True)
    fa, fb = pipeline.FusedHotPath(net_a.deconv_layers), pipeline.FusedHotPath(net_b.deconv_layers)
    # (global_range_active() asks torch.distributed for the world size: stand in for "two ranks whose extremes agree")
    saved = (pipeline.hotpath.global_range_active, pipeline.FusedHotPath._global_commit)      # (the name FusedHotPath's module resolves)

    def commit_local(act, dev, bits, mom, stream):
        st = act._device_state(dev)
        f = st.view(torch.float32)
        t = torch.stack((f[4], f[5]))
        N_.check(N_.lib().cdn_quantact_commit_range(act.x_min.data_ptr(), act.x_max.data_ptr(), st.data_ptr(),
                                                    t.data_ptr(), bits, mom, 1, stream), "commit")
    try:
        pipeline.hotpath.global_range_active = lambda acts: any(getattr(a, "global_range", False) for a in acts if a is not None)
        pipeline.FusedHotPath._global_commit = staticmethod(commit_local)
        for x in _batches(4, 3):
            ya, yb = fa(x.cuda()).clone(), fb(x.cuda()).clone()
            assert torch.equal(ya, yb)
        for (a_lo, a_hi), (b_lo, b_hi) in zip(_ranges(net_a), _ranges(net_b)):
            assert torch.equal(a_lo, b_lo) and torch.equal(a_hi, b_hi)
    finally:
        pipeline.hotpath.global_range_active, pipeline.FusedHotPath._global_commit = saved


def test_global_range_model_keeps_its_stages_on_the_fused_schedule():
    """enable_fused() on a model in global-range mode: backbone and heads go module by module (their fused schedules
    update ranges inside the producing kernels), the three deform stages stay fused (round 5) -- decided before any
    kernel runs, same detections-level outputs as the module path."""
    import copy
    import warnings
    from codenet_amd import harness, pipeline
    model = harness.create_model(quantize=True).cuda().eval()
    pipeline.set_global_range(model, True)
    m2 = copy.deepcopy(model).enable_fused()
    x = torch.randn(4, 3, 256, 256, generator=torch.Generator().manual_seed(3)).cuda()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = model(x)[-1]
        b = m2(x)[-1]
    assert m2._fpath is not None and m2._fheads is None
    for k in a:
        std = a[k].std().item() + 1e-6
        d = (a[k] - b[k]).abs()
        assert d.mean().item() < 0.12 * std, (k, d.mean().item(), std)


def _rccl_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)       # nccl == RCCL on ROCm
    from codenet_amd import pipeline
    seen = torch.ones(1, device="cuda")
    dist.all_reduce(seen)
    net = pipeline.build_hot_path(quantized=True).cuda()
    nbytes = pipeline.broadcast_parameters(net, src=0)
    t = torch.tensor([-1.5, 2.5], device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                            # the collective of FusedHotPath._global_commit
    dets = torch.arange(2 * 100 * 6, dtype=torch.float32, device="cuda").view(2, 100, 6)
    allg = pipeline.gather_detections(dets)
    # (the package's helpers return early on one rank: the same collectives called directly)
    flat = torch.randn(330_000, device="cuda")
    keep = flat.clone()
    dist.broadcast(flat, src=0)
    got = torch.empty(world * dets.numel(), device="cuda")
    dist.all_gather_into_tensor(got, dets.reshape(-1).contiguous())
    assert torch.equal(flat, keep) and torch.equal(got, dets.reshape(-1))
    torch.cuda.synchronize()
    torch.save({"seen": seen.item(), "nbytes": nbytes, "t": t.cpu(), "gathered": tuple(allg.shape)}, out % rank)
    dist.destroy_process_group()


def test_rccl_backend_runs_the_paths_collectives_on_one_rank(tmp_path):
    """The collectives of the multi-GPU path (start-up broadcast, detection all-gather, the global-range MAX all-reduce)
    on the RCCL backend itself, one rank on the box's one GPU: the backend loads, binds to the device and runs them on
    GPU tensors (gpurun gives one GPU; more ranks are the driver's round-end run)."""
    out = str(tmp_path / "n%d.pt")
    mp.spawn(_rccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    r = torch.load(out % 0)
    assert r["seen"] == 1.0 and r["nbytes"] == 0           # (one rank: broadcast_parameters has nothing to send)
    assert r["t"].tolist() == [-1.5, 2.5] and r["gathered"] == (2, 100, 6)
