"""Multi-process path on CPU (gloo, world_size 2): the start-up broadcast of parameters / buffers
and the image sharding used by bench.py --gpus N.  No GPU compute."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from codenet_amd import pipeline
    net = pipeline.build_hot_path(quantized=True, seed=317 if rank == 0 else 1000 + rank)
    for m in net.modules():                      # give QuantAct ranges rank-specific values
        if hasattr(m, "x_min"):
            m.x_min.fill_(-1.0 - rank)
            m.x_max.fill_(2.0 + rank)
    nbytes = pipeline.broadcast_parameters(net, src=0)
    sd = net.state_dict()
    digest = torch.stack([v.double().sum() for v in sd.values() if v.dtype.is_floating_point])
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    lo, hi = pipeline.shard_range(13, rank, world)
    torch.save({"same": all(torch.equal(gathered[0], g) for g in gathered), "nbytes": nbytes,
                "shard": (lo, hi), "xmin": float(sd["deconv_layers.0.quant_act.1.x_min"])},
               out % rank)
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2(tmp_path):
    world = 2
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = [torch.load(out % r) for r in range(world)]
    assert all(r["same"] for r in res)
    assert res[0]["nbytes"] == res[1]["nbytes"] > 1_000_000       # ~1.3 MB of fp32 state
    assert res[1]["xmin"] == -1.0                                  # rank 0's QuantAct range won
    assert [r["shard"] for r in res] == [(0, 7), (7, 13)]          # contiguous, sizes differ by <= 1


def _worker_global_range(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from codenet_amd import pipeline
    from codenet_amd.portable_quantizer.quant_modules import QuantAct
    torch.manual_seed(0)
    full = [torch.randn(6, 5, 4, 4) * (1.0 + 0.5 * i) + 0.3 * i for i in range(4)]       # the "one process" batches
    act = QuantAct(8, quant_mode="asymmetric")
    assert pipeline.set_global_range(act, True) == 1
    lo, hi = pipeline.shard_range(6, rank, world)
    outs = [act(x[lo:hi]) for x in full]
    torch.save({"outs": outs, "x_min": act.x_min.clone(), "x_max": act.x_max.clone(), "shard": (lo, hi)}, out % rank)
    dist.destroy_process_group()


def test_global_range_mode_equals_one_process_world2(tmp_path):
    """SURVEY 8(e)(3) parity mode: with QuantAct.global_range the batch extremes are all-reduced (MIN / MAX), so two
    ranks with 3 images each track EXACTLY the ranges -- and produce exactly the outputs -- of one process running
    all 6 images, over four forwards (the '+=' initialisation and three EMA steps)."""
    world = 2
    out = str(tmp_path / "g%d.pt")
    mp.spawn(_worker_global_range, args=(world, _free_port(), out), nprocs=world, join=True)
    res = [torch.load(out % r) for r in range(world)]
    sys.path.insert(0, ROOT)
    from codenet_amd.portable_quantizer.quant_modules import QuantAct
    torch.manual_seed(0)
    full = [torch.randn(6, 5, 4, 4) * (1.0 + 0.5 * i) + 0.3 * i for i in range(4)]
    ref = QuantAct(8, quant_mode="asymmetric")
    ref_outs = [ref(x) for x in full]
    for r in res:
        lo, hi = r["shard"]
        assert torch.equal(r["x_min"], ref.x_min) and torch.equal(r["x_max"], ref.x_max)
        for a, b in zip(r["outs"], ref_outs):
            assert torch.equal(a, b[lo:hi])
    # and without the mode the ranks' ranges differ from the one-process run (the mode is what makes them equal)
    own = QuantAct(8, quant_mode="asymmetric")
    for x in full:
        own(x[:3])
    assert not torch.equal(own.x_min, ref.x_min) or not torch.equal(own.x_max, ref.x_max)


def _worker_nan(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from codenet_amd import pipeline
    from codenet_amd.portable_quantizer.quant_modules import QuantAct, allreduce_extremes
    torch.manual_seed(rank)
    clean = allreduce_extremes(torch.tensor([-1.0 - rank]), torch.tensor([2.0 + rank]))
    act = QuantAct(8, quant_mode="asymmetric")
    pipeline.set_global_range(act, True)
    act(torch.randn(3, 5, 4, 4))
    before = (act.x_min.clone(), act.x_max.clone())
    x = torch.randn(3, 5, 4, 4)
    if rank == 1:
        x[1, 2, 3, 0] = float("nan")
    act(x)
    torch.save({"clean": clean, "before": before, "x_min": act.x_min.clone(), "x_max": act.x_max.clone()}, out % rank)
    dist.destroy_process_group()


def test_global_range_mode_propagates_a_nan_from_one_rank_world2(tmp_path):
    """gloo's (and RCCL's) MAX is not NaN-propagating, so a NaN in ONE rank's batch was dropped by the
    all-reduce while the one-process run's x.min() / x.max() would be NaN (quant_modules.py:203-219).  allreduce_extremes
    carries a flag: both ends are NaN on EVERY rank, and clean extremes still reduce to {min of mins, max of maxes}."""
    world = 2
    out = str(tmp_path / "n%d.pt")
    mp.spawn(_worker_nan, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in (torch.load(out % r) for r in range(world)):
        assert r["clean"].tolist() == [-2.0, 3.0]
        assert torch.isfinite(r["before"][0]).all() and torch.isfinite(r["before"][1]).all()
        assert torch.isnan(r["x_min"]).all() and torch.isnan(r["x_max"]).all()


def test_shard_range_partitions():
    from codenet_amd.pipeline import shard_range
    for total in (1, 7, 64, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _bench(*argv, env=None):
    import json
    import subprocess
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, capture_output=True,
                       text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(line[-1]) if line else None), p.stderr


def test_bench_launcher_starts_its_own_ranks_world2():
    """`python bench.py --gpus 2` run DIRECTLY (no torchrun around it) must start two ranks itself -- as child
    processes, before touching any GPU -- and report the number of ranks the collective saw.
    --plumbing-only --backend gloo: launcher, rendezvous, start-up broadcast, uneven image shards and the
    per-batch detections all_gather on CPU tensors; no kernels, no rate."""
    rc, res, err = _bench("--gpus", "2", "--plumbing-only", "--backend", "gloo", "--batch", "3")
    assert rc == 0, err[-2000:]
    assert res["n_gpus"] == 2 and res["replicas_identical"] and res["detections_in_rank_order"]
    assert res["detections_gathered"] == [7, 100, 6] and res["broadcast_bytes"] > 1_000_000
    assert res["value"] is None                                     # a plumbing run never reports a rate
    # shard sizes are exchanged ONCE at start-up; the per-batch gather is then one collective and no host read
    # (no count all_gather + .item() per batch)
    assert res["shard_sizes"] == [4, 3] and res["equal_shards_gathered"]
    assert res["host_syncs_in_gather"] == 0


def test_bench_refuses_world_size_mismatch():
    """--gpus N with a different WORLD_SIZE in the environment is an error (exit 2), not a silent 1-rank run."""
    rc, res, err = _bench("--gpus", "2", "--plumbing-only", "--backend", "gloo",
                          env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2 and res is None and "WORLD_SIZE 1" in err


def test_bench_presets():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse(["--config", "cfg4"])
    assert (a.w2, a.batch, a.res, a.fp32) == (True, 32, 512, False)          # 32 / GPU x 8 = 256 (configs[3])
    a = bench.parse([])
    assert (a.w2, a.batch, a.res, a.fp32) == (False, 64, 512, False)         # configs[2]
    a = bench.parse(["--config", "cfg2", "--batch", "8"])
    assert (a.w2, a.batch, a.res, a.fp32) == (False, 8, 256, True)
