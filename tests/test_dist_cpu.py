"""N > 1 path on CPU: reads sharded by rank (contiguous ranges of one global read stream), per-rank class-count
tables, one all-reduce (gloo here, RCCL on the GPUs). The per-rank mapper is the host lane emulator; the property under
test is that the reduced table equals the single-process table and is independent of the number of ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers


def shard_counts(rank, world, port, n_total, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pa = helpers.pa
    host = pa.build_index(str(helpers.FASTA), 24, 2)
    tx = pa.Txome.from_host_index(host)
    per = n_total // world
    first = rank * per                                         # bench.py: rank r owns reads [r*per, (r+1)*per)
    tiles, lens = tx.simulate_host(100, 3, per, 10000, first_read=first)
    r = helpers.Emu(host, 2).map_tiles(tiles, lens, 4)
    counts = torch.from_numpy(helpers.counts_reference(r["results"], r["coff"], r["ids"], host))
    dist.all_reduce(counts)                                    # the one exchange step of the path
    if rank == 0:
        ret.put(counts.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_sharded_counts_equal_single_process(built):
    n_total = 4000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=shard_counts, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    reduced = q.get(timeout=500)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    pa = helpers.pa
    host = pa.build_index(str(helpers.FASTA), 24, 4)
    tiles, lens = pa.Txome.from_host_index(host).simulate_host(100, 3, n_total, 10000)
    r = helpers.Emu(host).map_tiles(tiles, lens, 4)
    whole = helpers.counts_reference(r["results"], r["coff"], r["ids"], host)
    assert np.array_equal(reduced, whole) and reduced.sum() == n_total
