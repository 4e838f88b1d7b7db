"""Measurement helper (GPU box): where ShardedMagNetConv's construction spends its time at the north-star size when
`world` ranks share the one GPU over gloo (tests/test_gpu_fullsize.py) -- cProfile of rank 0, top entries by
cumulative time -> gpurun_out/sharded_init_profile.txt."""
import cProfile
import io
import os
import pstats
import socket
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rank_main(rank, world, port, path, layout):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if os.environ.get("PROBE_HW_QUEUES"):
        os.environ["GPU_MAX_HW_QUEUES"] = os.environ["PROBE_HW_QUEUES"]
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorch_geometric_signed_directed_amd.parallel import ShardedMagNetConv
    dev = torch.device("cuda:0")
    ei = torch.from_numpy(np.ascontiguousarray(np.load(path, mmap_mode="r"))).to(dev)
    torch.cuda.synchronize()
    prof = cProfile.Profile()
    t0 = time.perf_counter()
    prof.enable()
    layer = ShardedMagNetConv(64, 64, 1, 0.25, 1000000, ei, None, device=dev, layout=layout)
    torch.cuda.synchronize()
    prof.disable()
    dt = time.perf_counter() - t0
    if rank == 0:
        out = io.StringIO()
        pstats.Stats(prof, stream=out).sort_stats("cumulative").print_stats(45)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"sharded_init_profile_w{world}.txt"), "w") as fh:
            fh.write(f"world {world} layout {layout}: {dt:.2f} s\n" + out.getvalue())
        print(f"world {world}: {dt:.2f} s", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import bigdata
    path = bigdata.dsbm_graph(1000000, 20000000, seed=0)
    if os.environ.get("PROBE_PARENT_CTX"):          # the pytest process of the real tests holds a device context too
        if os.environ["PROBE_PARENT_CTX"] != "default":
            os.environ["GPU_MAX_HW_QUEUES"] = os.environ["PROBE_PARENT_CTX"]
        keep = torch.zeros(1 << 28, device="cuda:0")
        torch.cuda.synchronize()
        os.environ.pop("GPU_MAX_HW_QUEUES", None)
    for world in [int(a) for a in sys.argv[1:]] or [8]:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(rank_main, args=(world, port, path, "auto"), nprocs=world, join=True)
