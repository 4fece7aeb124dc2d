"""Worker process of tests/test_parallel.py (run as a script: `python _parallel_worker.py RANK WORLD PORT`)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tts_amd import parallel, synthetic  # noqa: E402


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = dict(upsample_initial_channel_decoder=32)
    sd = synthetic.make_vits_state(args, seed=9) if rank == 0 else None
    got = parallel.broadcast_state_dict(sd, src=0)
    ref = synthetic.make_vits_state(args, seed=9)
    ok = set(got) == set(ref) and all(torch.equal(got[k], ref[k]) and got[k].dtype == ref[k].dtype for k in ref)
    lo, hi = parallel.shard_range(33)
    dist.barrier()
    dist.destroy_process_group()
    print("RESULT " + json.dumps({"rank": rank, "ok": bool(ok), "lo": lo, "hi": hi}), flush=True)


if __name__ == "__main__":
    main()
