"""Builds the HIP library in-tree: ``python -m nfst_amd.build``.

One ``hipcc --offload-arch=gfx950`` command over ``csrc/pack.cpp``, ``csrc/chunk_pack.cpp`` (host
schedulers) and ``csrc/kernels.hip`` (C-ABI launchers; the kernels are in the
headers it includes: semiring, tile pipeline, forward-backward, path kernels) ->
``nfst_amd/lib/libnfst_hip.so``.  hipcc cross-compiles without a GPU.

Every build reads the compiler's per-kernel resource report (``-Rpass-analysis=kernel-resource-usage``),
keeps it as ``lib/libnfst_hip.resources.json`` and FAILS when a kernel breaks one of the register
assumptions the hand-written code relies on (``check_resources``).
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SRC = [os.path.join(CSRC, "pack.cpp"), os.path.join(CSRC, "chunk_pack.cpp"), os.path.join(CSRC, "kernels.hip")]
HDR = os.path.join(ROOT, "include", "nfst_hip.h")
OUT = os.path.join(HERE, "lib", "libnfst_hip.so")


def _deps():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]


def is_stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in SRC + _deps() + [HDR, os.path.abspath(__file__)])


_FIELDS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
           "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
           "LDS Size [bytes/block]": "lds"}


def parse_resources(text: str) -> dict:
    """{demangled-ish kernel name: {vgprs, agprs, sgpr_spill, vgpr_spill, occupancy, ...}} from hipcc's remarks."""
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass-analysis=kernel-resource-usage\]", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
            continue
        if cur is None or ":" not in body:
            continue
        k, v = body.rsplit(":", 1)
        if k.strip() in _FIELDS:
            try:
                cur[_FIELDS[k.strip()]] = int(v)
            except ValueError:
                pass
    return out


def _demangle(names):
    """``k_forward_backward<256, 0, true, false, false>`` from the mangled kernel names (template arguments of our
    kernels are ints and bools only; no external demangler needed)."""
    out = {}
    for n in names:
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
        if not m:
            out[n] = n
            continue
        ln = int(m.group(1))
        base, rest = n[m.end():m.end() + ln], n[m.end() + ln:]
        args = []
        if rest.startswith("I"):
            for t, v in re.findall(r"L([ibjm])(n?\d+)E", rest[1:rest.index("EE") + 1] if "EE" in rest else rest):
                args.append(("true" if v == "1" else "false") if t == "b" else v.replace("n", "-"))
        out[n] = base + ("<" + ", ".join(args) + ">" if args else "")
    return out


def check_resources(res: dict) -> list:
    """The register assumptions of the hand-written kernels:

    * the fused sweeps (``k_forward_backward<NT, 0, true, ...>``) stage tiles in the accumulation registers
      a0 .. a31 from inline asm: the compiler must have allocated exactly those 32 AGPRs (``AGPRs: 32``; more
      would mean it uses AGPRs itself -- it may then pick a0 .. a31 between two asm statements, the abort of
      round 2) and spilled no VGPR (its spills go to AGPRs first);
    * no sweep kernel may spill vector registers: the tile / sweep loops are latency chains.
    """
    bad = []
    for name, r in res.items():
        fused = re.search(r"k_forward_backward<\d+, 0, true", name) is not None
        if fused and (r.get("agprs") != 32 or r.get("vgpr_spill", 0) != 0):
            bad.append(f"{name}: fused sweep needs AGPRs == 32 and no VGPR spill, got AGPRs {r.get('agprs')}, "
                       f"VGPR spill {r.get('vgpr_spill')}")
        if re.search(r"k_(forward_backward|backward|viterbi_tw)<", name) and r.get("vgpr_spill", 0) != 0:
            bad.append(f"{name}: VGPR spill {r.get('vgpr_spill')} in a sweep kernel")
        if not fused and re.search(r"k_(forward_backward|backward)<", name) and r.get("agprs", 0) != 0:
            bad.append(f"{name}: {r.get('agprs')} AGPRs in a kernel that stages nothing in them")
    return bad


def build(force: bool = False, verbose: bool = False, variant: str = "", defines=()) -> str:
    """``variant`` + ``defines`` (-D flags): an experimental build of the same sources into
    lib/variants/libnfst_hip_<variant>.so (A/B runs inside one GPU box call; loaded through
    ``NFST_LIB`` when the package itself was imported with ``NFST_TUNING=1``)."""
    out = OUT if not variant else os.path.join(HERE, "lib", "variants", f"libnfst_hip_{variant}.so")
    if not variant and not force and not is_stale():
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread", "-std=c++17",
           "-I" + os.path.join(ROOT, "include"), "-Wall", "-Wextra", "-Wno-inline-asm", "-Wno-array-bounds",
           "-Rpass-analysis=kernel-resource-usage", *defines, *SRC, "-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    noise = "-Rpass-analysis=kernel-resource-usage]"
    diag = "\n".join(l for l in r.stderr.splitlines() if noise not in l)
    if r.returncode != 0:
        sys.stderr.write(diag + "\n")
        raise subprocess.CalledProcessError(r.returncode, cmd)
    if diag.strip() and verbose:
        sys.stderr.write(diag + "\n")
    res = parse_resources(r.stderr)
    names = _demangle(list(res))
    res = {names[k]: v for k, v in res.items()}
    bad = check_resources(res)
    if bad:
        os.remove(out + ".tmp")
        raise RuntimeError("register assumptions of the hand-written kernels do not hold:\n  " + "\n  ".join(bad))
    os.replace(out + ".tmp", out)
    with open(out[:-3] + ".resources.json", "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    if verbose:
        for k in sorted(res):
            v = res[k]
            print(f"{v.get('vgprs', 0):4d} VGPR {v.get('agprs', 0):3d} AGPR  spill s{v.get('sgpr_spill', 0)} v{v.get('vgpr_spill', 0)}  occ {v.get('occupancy', 0)}  {k.split('(')[0]}")
    return out


if __name__ == "__main__":
    variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, variant=variant,
                defines=[a for a in sys.argv[1:] if a.startswith("-D")]))
