"""`-m gpu` tier, the N > 1 path on real GPUs: self-checking for the day a multi-GPU node runs this suite.
The one-rank forms run on every GPU box (they execute the product's RCCL calls with a communicator of one); the two-rank forms
need two GPUs and skip otherwise."""
import json
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

import helpers

pa = helpers.pa
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(nproc, script, *args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)] + [str(a) for a in args]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(helpers.ROOT))


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("ranks", [1, 2])
def test_product_collectives_reduce_to_the_one_gpu_tables(ranks):
    """tests/dist_gpu_worker.py under torchrun: reads sharded by rank, pa_counts_allreduce + pa_overflow_allgather (the product's
    RCCL communicator), reduced table and merged overflow == the one-GPU results of the same global read range"""
    if _gpus() < ranks:
        pytest.skip("needs %d GPUs" % ranks)
    out = _torchrun(ranks, helpers.ROOT / "tests" / "dist_gpu_worker.py", 400_000)
    assert out.returncode == 0 and "DIST-OK ranks=%d" % ranks in out.stdout and "rccl_ranks=%d" % ranks in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize("ranks", [2])
def test_bench_under_torchrun_reduces_through_the_product(ranks):
    """bench.py --gpus 2 as the driver launches it: the line must say the reduce went through pa_counts_allreduce (rccl_ranks == 2,
    not the torch.distributed spare) and count every read of every rank"""
    if _gpus() < ranks:
        pytest.skip("needs %d GPUs" % ranks)
    out = _torchrun(ranks, helpers.ROOT / "bench.py", "--gpus", ranks, "--steps", 2, "--warmup", 1, "--workload", "config2", "--batch", 2_000_000)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == ranks and line["rccl_ranks"] == ranks and "pa_counts_allreduce" in line["config"]["parallelism"]
    assert line["scaling"] == "weak" and line["parity_sample"]["bit_exact_vs_oracle"] is True


def test_bench_two_ranks_on_one_gpu_rehearsal():
    """bench.py --gpus 2 with both ranks on THIS box's one GPU (PA_BENCH_BACKEND=gloo: RCCL refuses two ranks on a device, so the count tables are reduced
    through torch.distributed's gloo spare and the line says so: rccl_ranks 0): the sharding of the reads by rank, the barrier + max-over-ranks timing, the
    gather of every rank's kernel times and the whole-job value are the code the driver's N > 1 runs take"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PA_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(helpers.ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "config2", "--batch", "2000000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(helpers.ROOT))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 0 and line["scaling"] == "weak"
    assert line["per_rank"]["ranks"] == 2 and len(line["per_rank"]["kernel_ms"]) == 2 and min(line["per_rank"]["kernel_ms"]) > 0
    assert abs(line["value"] - 2 * 2 * 2_000_000 / (line["ms_per_step"] * 2 / 1000.0)) < 1e-6 * line["value"]   # both ranks' reads over the slower rank's time
    assert line["parity_sample"]["bit_exact_vs_oracle"] is True


def test_index_create_multi_one_thread_per_gpu(small_index):
    """pa_index_create_multi(ndev = all GPUs of the box) driven from ONE process with one thread per GPU: the shards' tables add up
    to the one-GPU table of the whole range (ndev = 1 on a one-GPU box: the same code path, one handle)"""
    import ctypes as C
    import torch
    ndev = min(_gpus(), 4)
    host = small_index(24)
    flat = host.flat()
    devices = (C.c_int * ndev)(*range(ndev))
    handles = (pa._ffi.vp * ndev)()
    pa.check(pa.lib().pa_index_create_multi(C.byref(flat), devices, ndev, handles))
    try:
        tx = pa.Txome.from_host_index(host)
        n_total, wpr = 300_000, 4
        per = n_total // ndev
        tables, errors = [None] * ndev, []

        def shard(d):
            try:
                dev = torch.device("cuda", d)
                with torch.cuda.device(dev):
                    h = handles[d]
                    tiles = torch.zeros(pa.lib().pa_tiles_words(per, wpr), dtype=torch.int64, device=dev)
                    lens = torch.zeros(per, dtype=torch.int32, device=dev)
                    st = torch.cuda.Stream(device=dev)
                    tx.simulate_device(100, 5, per, tiles.data_ptr(), lens.data_ptr(), 10000, d * per, wpr, d, st.cuda_stream)
                    cap = pa.lib().pa_map_arena_hint(h, per)
                    res = torch.zeros(per * 4, dtype=torch.int32, device=dev)
                    arena = torch.zeros(cap, dtype=torch.int32, device=dev)
                    counts = torch.zeros(pa.lib().pa_counts_len(h), dtype=torch.int64, device=dev)
                    pa.check(pa.lib().pa_map_count_batch_device(h, tiles.data_ptr(), lens.data_ptr(), per, wpr, 2, res.data_ptr(), arena.data_ptr(), cap,
                                                                counts.data_ptr(), st.cuda_stream))
                    used, need = C.c_uint64(), C.c_uint64()
                    pa.check(pa.lib().pa_map_finish(h, st.cuda_stream, C.byref(used), C.byref(need)))
                    pa.check(pa.lib().pa_index_release_stream(h, st.cuda_stream))
                    tables[d] = counts.cpu().numpy()
            except Exception as e:   # noqa: BLE001
                errors.append(repr(e))
        threads = [threading.Thread(target=shard, args=(d,)) for d in range(ndev)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        total = np.sum(tables, axis=0)
        h_tiles, h_lens = tx.simulate_host(100, 5, per * ndev, 10000, 0, wpr)
        o_res, o_coff, o_ids, _ = helpers.Oracle(host).map_tiles(h_tiles, h_lens, wpr, 2, 8)
        want = helpers.counts_reference_fast(o_res, o_coff, o_ids, host)
        assert np.array_equal(total.astype(np.int64), want.astype(np.int64)) and int(total.sum()) == per * ndev
    finally:
        for d in range(ndev):
            if handles[d]:
                pa.lib().pa_index_destroy(handles[d])
