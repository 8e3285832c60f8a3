tag=r5w
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "process_reads or record_stream or c_client or fastq" 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
for w in 1 300 5000 100000; do echo "window $w:"; PA_INGEST_WINDOW=$w python -m pytest tests -m gpu -x -q -k "process_reads or fastq" 2>&1 | grep -E "passed|failed|rror|assert" | tail -4; PA_INGEST_WINDOW=$w python tools/gpu_fastq_fuzz.py 150 2>&1 | tail -2; done
python tools/gpu_fastq_fuzz.py 200 2>&1 | tail -2
for i in 1 2 3; do python tools/bench_ingest.py --reads 8000000 --threads 16,16 2>&1 | grep -E "pa ingest\] 8|value" | cut -c1-330; done
for w in 67108864 134217728 536870912; do echo "window $w"; PA_INGEST_WINDOW=$w python tools/bench_ingest.py --reads 8000000 --threads 16,16 2>&1 | grep -E "pa ingest\] 8|value" | cut -c1-330; done
