"""Static check of filter_kernel's hand-issued vector-memory pipeline (vidtome_amd/csrc/match_filter.hip).

The kernel issues its fragment loads and LDS-DMA pieces through inline asm and waits with hand-counted
`s_waitcnt vmcnt(N)`.  The compiler does not know that the asm outputs are still in flight, so this test compiles
the file to assembly and verifies, on the emitted code, that
  * no instruction touches a register that an asm load is still filling before one of the counted waits, and
  * the compiler itself never reads m0, which the asm LDS-DMA statements overwrite.
Needs hipcc only (cross-compiles without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "vidtome_amd", "csrc", "match_filter.hip")


def _regs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(r) for r in re.findall(r"\bv(\d+)\b", text))
    return out


@pytest.fixture(scope="module")
def filter_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from vidtome_amd import build
    out = tmp_path_factory.mktemp("asm") / "match_filter.s"
    flags = [f for f in build.FLAGS if f not in ("-fPIC",)]
    subprocess.run([hipcc, *flags, "-S", "--cuda-device-only", SRC, "-o", str(out)], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    lines = out.read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if "filter_kernel" in l and l.rstrip().endswith(":") is False and l.startswith("_ZN"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


def test_asm_loads_are_awaited_before_use(filter_asm):
    lines = filter_asm
    in_asm = False
    loads = []          # (line index, destination registers)
    for i, l in enumerate(lines):
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif in_asm and "global_load_dwordx4" in l and "lds" not in l:
            dst = l.split("global_load_dwordx4")[1].split(",")[0]
            loads.append((i, _regs(dst)))
    assert len(loads) >= 8, "the hand-issued fragment loads were not found"
    for i, dst in loads:
        waited = False
        for j in range(i + 1, len(lines)):
            l = lines[j].split(";")[0]
            if "s_waitcnt" in l and "vmcnt" in l:
                waited = True
            if l.strip().startswith((".", "s_branch", "s_cbranch", "s_endpgm")) and not waited and "s_cbranch" in l:
                continue
            if _regs(l) & dst and "global_load_dwordx4" not in l:
                assert waited, f"line {j} uses {sorted(_regs(l) & dst)} of the load at line {i} before any vmcnt wait:\n{lines[j]}"
                break


def test_compiler_does_not_touch_m0(filter_asm):
    in_asm = False
    for l in filter_asm:
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif not in_asm and re.search(r"\bm0\b", l.split(";")[0]):
            raise AssertionError("compiler-generated instruction uses m0, which the asm LDS-DMA overwrites: " + l)
