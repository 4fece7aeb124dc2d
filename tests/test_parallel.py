"""Multi-process (gloo, world_size 2, CPU) test of the only multi-GPU mechanism the path needs: the one-shot
flat-blob weight broadcast + utterance sharding (tts_amd/parallel.py)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from tts_amd import parallel, synthetic

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = dict(upsample_initial_channel_decoder=32)
    sd = synthetic.make_vits_state(args, seed=9) if rank == 0 else None
    got = parallel.broadcast_state_dict(sd, src=0)
    ref = synthetic.make_vits_state(args, seed=9)
    ok = set(got) == set(ref) and all(torch.equal(got[k], ref[k]) and got[k].dtype == ref[k].dtype for k in ref)
    lo, hi = parallel.shard_range(33)
    q.put((rank, ok, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_shard_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], "broadcast state_dict differs from the source"
    assert (res[0][2], res[0][3], res[1][2], res[1][3]) == (0, 17, 17, 33)


def test_flatten_roundtrip_and_length_sharding():
    sys.path.insert(0, ROOT)
    from tts_amd import parallel

    sd = {"a.weight": torch.randn(3, 4, 5), "b.bias": torch.randn(7), "n": torch.tensor([3], dtype=torch.int64)}
    blob, man = parallel.flatten_state_dict(sd)
    back = parallel.unflatten_state_dict(blob, man)
    assert all(torch.equal(back[k], sd[k]) and back[k].dtype == sd[k].dtype for k in sd)
    lens = [10, 50, 20, 40, 30, 60, 5]
    shards = parallel.shard_by_length(lens, 3)
    assert sorted(sum(shards, [])) == list(range(7)) and max(map(len, shards)) - min(map(len, shards)) <= 1
    tot = [sum(lens[i] for i in s) for s in shards]
    assert max(tot) - min(tot) <= max(lens)
