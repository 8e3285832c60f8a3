"""gcc recipe for the CPU oracle (checker only; never part of the product)."""
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
ORACLE_SO = HERE / "_build" / "libpa_oracle.so"


def build_oracle(force: bool = False) -> Path:
    srcs = [HERE / "pa_oracle.c", HERE / "pa_oracle.h"]
    if force or not ORACLE_SO.exists() or any(s.stat().st_mtime > ORACLE_SO.stat().st_mtime for s in srcs):
        ORACLE_SO.parent.mkdir(parents=True, exist_ok=True)
        cmd = ["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-pthread", "-Wall", str(srcs[0]), "-o", str(ORACLE_SO)]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + proc.stderr)
    return ORACLE_SO


if __name__ == "__main__":
    print(build_oracle(True))
