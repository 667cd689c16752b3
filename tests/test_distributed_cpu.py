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


def test_shard_range_partitions():
    from codenet_amd.pipeline import shard_range
    for total in (1, 7, 64, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
