"""Reads the gfx950 code objects out of fcc_qp_amd/libfccqp_hip.so (test infrastructure).

hipcc embeds one clang offload bundle per translation unit in the section `.hip_fatbin`; every
bundle holds a host placeholder and the gfx950 ELF.  The ELF's note record carries the per-kernel
resource usage (registers, scratch, LDS) and `llvm-objdump -d` gives the shipped instruction stream:
both are what tests/test_code_objects.py checks (no scratch in the fused kernels, and the hand-placed
VALU-write -> DPP-read wait states the compiler's hazard recogniser cannot see inside inline asm).
Works in the build container: nothing here needs a GPU.
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import tempfile
from typing import Dict, List, Tuple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "fcc_qp_amd", "libfccqp_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _tool(name: str) -> str:
    path = os.path.join(LLVM, name)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found")
    return path


def extract(lib: str = LIB, outdir: str | None = None) -> List[str]:
    """Writes every gfx950 code object of `lib` to outdir; returns the paths."""
    outdir = outdir or tempfile.mkdtemp(prefix="fccqp_co_")
    fat = os.path.join(outdir, "fatbin.bin")
    subprocess.check_call([_tool("llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    data = open(fat, "rb").read()
    paths = []
    pos = data.find(MAGIC)
    while pos >= 0:
        (count,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
        cur = pos + len(MAGIC) + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", data, cur)
            triple = data[cur + 24:cur + 24 + tlen].decode()
            cur += 24 + tlen
            if "gfx950" in triple and size:
                path = os.path.join(outdir, f"co{len(paths)}.elf")
                with open(path, "wb") as f:
                    f.write(data[pos + off:pos + off + size])
                paths.append(path)
        pos = data.find(MAGIC, pos + len(MAGIC))
    return paths


def kernel_metadata(elf: str) -> Dict[str, Dict[str, int]]:
    """{mangled kernel name: the kernel's record of the AMDGPU metadata note (vgpr_count, agpr_count,
    sgpr_count, private_segment_fixed_size, group_segment_fixed_size, *_spill_count, ...)}"""
    import yaml

    txt = subprocess.check_output([_tool("llvm-readelf"), "--notes", elf], text=True)
    start = txt.index("---")
    end = txt.index("\n...", start) if "\n..." in txt[start:] else len(txt)
    meta = yaml.safe_load(txt[start:end])
    return {k[".name"]: {key.lstrip("."): val for key, val in k.items() if not isinstance(val, (list, dict))}
            for k in meta.get("amdhsa.kernels", [])}


def demangle(names: List[str]) -> Dict[str, str]:
    import shutil
    filt = shutil.which("c++filt") or _tool("llvm-cxxfilt")
    txt = subprocess.check_output([filt] + names, text=True).splitlines()
    return dict(zip(names, txt))


def disassemble(elf: str) -> Dict[str, List[Tuple[str, str]]]:
    """{symbol: [(mnemonic, operand text), ...]} for every function of the code object."""
    txt = subprocess.check_output([_tool("llvm-objdump"), "-d", "--mcpu=gfx950", "--no-show-raw-insn", elf], text=True)
    funcs: Dict[str, List[Tuple[str, str]]] = {}
    cur = None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        parts = body.split(None, 1)
        cur.append((parts[0], parts[1] if len(parts) > 1 else ""))
    return funcs
