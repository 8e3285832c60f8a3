#!/usr/bin/env python3
"""Host link of the box: pinned host <-> HBM copy rates, one direction at a time and both at once (the practical ceiling of
bench.py's host-to-host leg; DESIGN.md §4 prices that leg against 63 GB/s, the nominal PCIe Gen5 x16 rate)."""
import json, time, torch

def main():
    n = 1 << 30
    h_a = torch.empty(n, dtype=torch.uint8).pin_memory()
    h_b = torch.empty(n, dtype=torch.uint8).pin_memory()
    d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
    d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def run(h2d, d2h, reps=8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if h2d:
                with torch.cuda.stream(s1): d_a.copy_(h_a, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2): h_b.copy_(d_b, non_blocking=True)
        torch.cuda.synchronize()
        return reps * n / (time.perf_counter() - t0) / 1e9
    run(True, True, 2)
    out = {"h2d_GBps": run(True, False), "d2h_GBps": run(False, True), "both_each_GBps": run(True, True), "bytes_per_copy": n}
    print(json.dumps(out))

if __name__ == "__main__":
    main()
