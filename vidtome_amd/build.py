"""Build libvidtome_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m vidtome_amd.build [--force]

The library is built IN-TREE (vidtome_amd/lib/) so that it travels with the repository snapshot to the
GPU box; it is git-ignored.  No fast-math: the matching path's fp32 arithmetic is part of the bit-exact
contract (IEEE divide / sqrt, explicit fma only).
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libvidtome_hip.so")
SOURCES = ["api.hip", "normalize.hip", "match.hip", "match_filter.hip", "sort.hip", "order.hip", "plan.hip", "gather.hip", "attention.hip", "attention16.hip", "attention16g.hip", "ddim.hip", "layernorm.hip", "geglu.hip", "linear.hip", "ff.hip"]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
         # MFMA results straight into VGPRs (gfx950 has a unified file): no v_accvgpr_read/write traffic
         # between the matrix results and the VALU softmax / running-max code
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, defines=(), tag: str = "") -> str:
    """Build the library.  `defines` / `tag` produce an experiment variant next to the shipped one
    (lib/variants/<tag>/libvidtome_hip.so, selected with VIDTOME_HIP_LIB; used by tools/kbench.py comparisons)."""
    libdir = os.path.join(LIBDIR, "variants", tag) if tag else LIBDIR
    lib = os.path.join(libdir, "libvidtome_hip.so")
    os.makedirs(libdir, exist_ok=True)
    cc = hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "ablate.h"), os.path.join(HERE, "..", "include", "vidtome_hip.h")]
    objs, jobs = [], []
    flags = FLAGS + ["-D" + d for d in defines]
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(libdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([cc, *flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + p.stdout)
        return p.stdout

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for out in ex.map(run, jobs):
            if verbose and out.strip():
                print(out)
    if force or jobs or _stale(lib, objs):
        run([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib, *objs])
    return lib


def kernel_resources(obj: str) -> dict:
    """Register / scratch budget of every gfx950 kernel in one of the built objects (`lib/<source>.o`), from the code
    object's metadata: {demangled-ish name: {vgpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count,
    private_segment_fixed_size}}.  tests/test_host.py pins the hot kernels with it (the filter's hand-counted memory
    pipeline must not spill; the attention kernel must keep 4 waves per SIMD)."""
    import re
    import tempfile
    llvm = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc()))), "lib", "llvm", "bin")
    if not os.path.isdir(llvm):
        llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as tmp:
        fat, dev = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        for cmd in ([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj],
                    [os.path.join(llvm, "clang-offload-bundler"), "--type=o", "--unbundle", "--input=" + fat, "--output=" + dev,
                     "--targets=hipv4-amdgcn-amd-amdhsa--" + ARCH]):
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if p.returncode != 0:
                raise RuntimeError(" ".join(cmd) + "\n" + p.stdout)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", dev], stdout=subprocess.PIPE, text=True,
                               check=True).stdout
    out, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\S+)", line)
        if not m:
            continue
        if m.group(1) == "name":
            cur = out.setdefault(m.group(2), {})
        elif cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return {k: v for k, v in out.items() if "vgpr_count" in v}


if __name__ == "__main__":
    # python -m vidtome_amd.build [--force] [--tag NAME -DMACRO ...]
    argv = sys.argv[1:]
    tag = argv[argv.index("--tag") + 1] if "--tag" in argv else ""
    print(build(force="--force" in argv, verbose=True, defines=[a[2:] for a in argv if a.startswith("-D")], tag=tag))
