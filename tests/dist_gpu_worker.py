"""One rank of the multi-GPU self-check (launched under torch.distributed.run by tests/test_gpu_dist.py, one process per GPU).

Every rank maps its contiguous share of ONE global read range with the class-count table and an overflow table attached, then
the product's own collectives reduce them: pa_counts_allreduce (RCCL all-reduce of the dense table) and pa_overflow_allgather
(RCCL all-gather + merge by content of the novel classes). Rank 0 then maps the WHOLE range on its own GPU and compares: the
reduced table and the merged overflow table must equal the one-GPU results, whatever the number of ranks. Prints one line,
"DIST-OK ranks=<n> ..." on success."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import helpers  # noqa: E402

pa = helpers.pa


def map_range(aligner, tx, first, n, wpr, dev, ovf):
    tiles = torch.zeros(pa.lib().pa_tiles_words(n, wpr), dtype=torch.int64, device=dev)
    lens = torch.zeros(n, dtype=torch.int32, device=dev)
    tx.simulate_device(100, 11, n, tiles.data_ptr(), lens.data_ptr(), 30000, first, wpr, dev.index)   # 3 % substitutions: novel classes
    cap = aligner.arena_hint(n)
    res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    arena = torch.zeros(cap, dtype=torch.int32, device=dev)
    counts = torch.zeros(aligner.counts_len(), dtype=torch.int64, device=dev)
    aligner.set_overflow(ovf)
    aligner.map_count_batch_device(tiles.data_ptr(), lens.data_ptr(), n, wpr, res.data_ptr(), arena.data_ptr(), cap, counts.data_ptr(), 2)
    aligner.map_finish()
    aligner.set_overflow(None)
    return counts


def main():
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    box = [pa.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)                    # torch.distributed is only the bootstrap of the 128-byte id
    comm = pa.Comm(local, world, rank, box[0])                # the PRODUCT's communicator (RCCL, owned by the library)
    assert (comm.rank, comm.size) == (rank, world)
    host = pa.build_index(str(helpers.FASTA), 24, 4)
    aligner = pa.Pseudoaligner(host, local)
    tx = pa.Txome.from_host_index(host)
    wpr = 4
    per = n_total // world
    ovf = pa.Overflow(local, 1 << 16, 1 << 22)
    counts = map_range(aligner, tx, rank * per, per, wpr, dev, ovf)
    aligner.counts_allreduce(counts.data_ptr(), comm)         # pa_counts_allreduce
    torch.cuda.synchronize()
    merged = ovf.allgather(comm)                              # pa_overflow_allgather
    reduced = counts.cpu().numpy()
    ok = True
    if rank == 0:
        ovf1 = pa.Overflow(local, 1 << 16, 1 << 22)
        whole = map_range(aligner, tx, 0, per * world, wpr, dev, ovf1).cpu().numpy()
        one = ovf1.fetch()
        ok = bool(np.array_equal(reduced, whole)) and int(reduced.sum()) == per * world and bool(np.array_equal(merged, one))
        novel = pa.parse_overflow(merged)
        ok = ok and sum(novel.values()) == int(whole[-3]) and len(novel) > 0
        print("DIST-%s ranks=%d reads=%d novel_classes=%d rccl_ranks=%d" % ("OK" if ok else "MISMATCH", world, per * world, len(novel), comm.size), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
