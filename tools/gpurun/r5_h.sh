tag=r5h
python -m pytest tests -m gpu -x -q -k "process_reads or record_stream or c_client or fastq" 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > gpurun_out/${tag}_pytest.txt; cat gpurun_out/${tag}_pytest.txt
python tools/gpu_fastq_fuzz.py 200 2>&1 | tail -3
for i in 1 2 3; do python tools/bench_ingest.py --reads 8000000 2>/dev/null | tail -4; done
