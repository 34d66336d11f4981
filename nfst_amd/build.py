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


def build(force: bool = False, verbose: bool = False, variant: str = "", defines=()) -> str:
    """``variant`` + ``defines`` (-D flags): an experimental build of the same sources into
    lib/variants/libnfst_hip_<variant>.so, loaded instead of the product library when the environment
    names it in NFST_LIB (profiles/tune/*.sh: A/B runs inside one GPU box call)."""
    out = OUT if not variant else os.path.join(HERE, "lib", "variants", f"libnfst_hip_{variant}.so")
    if not variant and not force and not is_stale():
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread", "-std=c++17",
           "-I" + os.path.join(ROOT, "include"), "-Wall", "-Wextra", "-Wno-inline-asm", *defines, *SRC, "-o", out + ".tmp"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, variant=variant,
                defines=[a for a in sys.argv[1:] if a.startswith("-D")]))
