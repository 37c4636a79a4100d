"""Builds libsph_mi355x.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

Flags that are part of the arithmetic contract (see csrc/sph_device.h):
  -ffp-contract=off                          no FMA contraction: products and sums round separately
  -fhip-fp32-correctly-rounded-divide-sqrt   IEEE f32 divide / sqrt on the device
  (no -ffast-math, denormals kept)
and one that is about speed only:
  -fno-slp-vectorize   left on, the SLP vectoriser pairs scalar f32 operations into v_pk_mul/add/fma_f32 plus the v_mov shuffles that
                       feed them; on gfx950 a packed f32 instruction issues in 4 cycles against 2 for each scalar one (tools/valu_issue.hip),
                       so the pairs gain nothing and the moves cost: the pair body of the sweeps is 12 % faster without it
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsph_mi355x.so")
SOURCES = ["sph_mi355x.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "sph_mi355x.h")]

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function",
]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libsph_mi355x.so cannot be built (there is no CPU fallback)")


def sources_sha256():
    """One digest of what the kernels are built from: every file of csrc/, the public header and the compiler flags.  Measurements that are
    committed and quoted later (profiles/pmc_traffic.json, profiles/valu_mix.json) carry it, and bench.py labels them "stale" when it differs
    from the sources it runs (tools/pmc_traffic.py, tools/valu_mix.py, bench.py)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(os.path.basename(name).encode() + b"\0" + f.read() + b"\0")
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + FLAGS + list(extra) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra=["-Rpass-analysis=kernel-resource-usage"] if "--resources" in sys.argv else [])
    print(LIB)
