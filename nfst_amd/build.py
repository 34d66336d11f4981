"""Builds the HIP library in-tree: ``python -m nfst_amd.build``.

One ``hipcc --offload-arch=gfx950`` command over ``csrc/pack.cpp`` (host
scheduler) and ``csrc/kernels.hip`` (C-ABI launchers; the kernels are in the
headers it includes: semiring, tile pipeline, forward-backward, path kernels) ->
``nfst_amd/lib/libnfst_hip.so``.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "pack.cpp"), os.path.join(HERE, "csrc", "kernels.hip")]
HDR = os.path.join(ROOT, "include", "nfst_hip.h")
DEPS = [os.path.join(HERE, "csrc", f) for f in ("semiring.h", "tile_pipeline.h", "fb_kernels.h", "path_kernels.h", "neural_kernels.h")]
OUT = os.path.join(HERE, "lib", "libnfst_hip.so")


def is_stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in SRC + DEPS + [HDR])


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread", "-std=c++17",
           "-I" + os.path.join(ROOT, "include"), "-Wall", "-Wextra", "-Wno-inline-asm", *SRC, "-o", OUT + ".tmp"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
