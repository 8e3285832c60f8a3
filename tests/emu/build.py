"""g++ recipe for the host lane emulator (checker for the CPU-only test tier; never part of the product)."""
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CSRC = ROOT / "rust-pseudoaligner_amd" / "csrc"
EMU_SO = HERE / "_build" / "libpa_emu.so"


def build_emu(force: bool = False) -> Path:
    srcs = [HERE / "emu_map.cpp"] + [CSRC / s for s in ("host_index.cpp", "dbg_build.cpp", "device_flatten.cpp")]
    deps = srcs + list(CSRC.glob("*.hpp")) + [HERE / "emu_loop.inc"] + [ROOT / "include" / "pseudoaligner_amd.h"]
    if force or not EMU_SO.exists() or any(s.stat().st_mtime > EMU_SO.stat().st_mtime for s in deps):
        EMU_SO.parent.mkdir(parents=True, exist_ok=True)
        # (-DPA_DEBUG_KNOBS: the emulator's flattener honours PA_DICT_LOAD, so that the tests can build DENSE dictionaries — keys in other
        # slots of their bucket, in the next buckets — which the shipped library's sparse table hardly ever produces)
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wno-unused-function", "-DPA_DEBUG_KNOBS"] + [str(s) for s in srcs] + ["-o", str(EMU_SO)]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError("emulator build failed:\n" + proc.stderr)
    return EMU_SO


if __name__ == "__main__":
    print(build_emu(True))
