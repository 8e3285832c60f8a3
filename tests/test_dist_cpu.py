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
    # ... and its second half: the novel classes (one slot of the dense table) travel as serialised overflow tables,
    # all-gathered at a common size and merged by content on every rank (pa_overflow_allgather does this over RCCL)
    mine = pa.serialise_overflow(helpers.novel_reference(r["results"], r["coff"], r["ids"], host))
    size = torch.tensor([len(mine)])
    dist.all_reduce(size, op=dist.ReduceOp.MAX)
    padded = torch.zeros(int(size.item()), dtype=torch.int64)
    padded[: len(mine)] = torch.from_numpy(mine.astype(np.int64))
    gathered = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded)
    merged = pa.overflow_merge([g.numpy().astype(np.uint32) for g in gathered])
    if rank == 0:
        ret.put((counts.numpy().copy(), merged))
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
    reduced, merged = q.get(timeout=500)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    pa = helpers.pa
    host = pa.build_index(str(helpers.FASTA), 24, 4)
    tiles, lens = pa.Txome.from_host_index(host).simulate_host(100, 3, n_total, 10000)
    r = helpers.Emu(host).map_tiles(tiles, lens, 4)
    whole = helpers.counts_reference(r["results"], r["coff"], r["ids"], host)
    assert np.array_equal(reduced, whole) and reduced.sum() == n_total
    novel = helpers.novel_reference(r["results"], r["coff"], r["ids"], host)
    assert pa.parse_overflow(merged) == novel and sum(novel.values()) == whole[-3] and len(novel) > 0
    assert np.array_equal(merged, pa.serialise_overflow(novel))                   # canonical: independent of the rank count


def test_overflow_merge_semantics(built):
    pa = helpers.pa
    a = pa.serialise_overflow({(1, 2, 3): 5, (7,): 2})
    b = pa.serialise_overflow({(1, 2, 3): 1 << 33, (9, 10): 4})
    padded = np.concatenate([b, np.zeros(7, np.uint32)])                          # gathered buffers are zero padded to a common size
    m = pa.overflow_merge([a, padded, np.zeros(0, np.uint32), np.zeros(5, np.uint32)])
    assert pa.parse_overflow(m) == {(1, 2, 3): (1 << 33) + 5, (7,): 2, (9, 10): 4}
    assert np.array_equal(m, pa.overflow_merge([padded, a]))                      # order of the ranks does not matter
    assert np.array_equal(pa.overflow_merge([m]), m)                              # idempotent
    assert pa.parse_overflow(pa.overflow_merge([])) == {}
    bad = a.copy()
    bad[2] = 1000                                                                 # a record that runs past the end
    with pytest.raises(pa.PaError):
        pa.overflow_merge([bad])
