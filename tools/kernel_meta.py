#!/usr/bin/env python3
"""Register / scratch metadata of the gfx950 kernels of one csrc/*.hip file, read from the compiler's own AMDHSA metadata
(`hipcc --cuda-device-only -S`, the flags of csrc/Makefile): name, VGPRs, AGPRs, VGPR / SGPR spill counts, scratch bytes.

The hand-scheduled kernels (gemm_quad.hip, gemm_narrow.hip) count their `s_waitcnt vmcnt(N)` by hand: a compiler that spilled, or sank a
plain load into their K loop, would make those counts wrong without any signal -- tests/test_host_logic.py asserts 0 spills / 0 scratch
for them on every build.   usage: python tools/kernel_meta.py teochat_amd/csrc/gemm_quad.hip [substring]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=16", "--cuda-device-only", "-S"]


def kernel_meta(src):
    """[{name, vgpr_count, agpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, ...}] for every kernel of `src`."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run([HIPCC] + FLAGS + ["-I" + os.path.join(ROOT, "include"), os.path.abspath(src), "-o", out], capture_output=True, text=True,
                           cwd=os.path.dirname(os.path.abspath(src)))
        if r.returncode != 0:
            raise RuntimeError("hipcc -S failed:\n" + r.stderr[-3000:])
        text = open(out).read()
    m = re.search(r"amdhsa\.kernels:(.*?)amdhsa\.target:", text, re.S)
    if not m:
        raise RuntimeError("no amdhsa.kernels metadata in the assembly")
    kernels = []
    for blk in re.split(r"\n  - \.agpr_count:", "\n" + m.group(1))[1:]:
        blk = "    .agpr_count:" + blk
        d = {}
        for key in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                    "group_segment_fixed_size", "max_flat_workgroup_size"):
            mm = re.search(r"\." + key + r":\s+(\d+)", blk)
            d[key] = int(mm.group(1)) if mm else None
        mm = re.search(r"\n    \.name:\s+(\S+)", blk)
        d["mangled"] = mm.group(1) if mm else "?"
        kernels.append(d)
    import shutil
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    names = (subprocess.run([filt] + [k["mangled"] for k in kernels], capture_output=True, text=True).stdout.split("\n")
             if filt else [k["mangled"] for k in kernels])
    for k, n in zip(kernels, names):
        k["name"] = re.sub(r"\(.*$", "", n.replace("void teo::", "").replace("unsigned short", "bf16"))
    return kernels


if __name__ == "__main__":
    ks = kernel_meta(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in ks:
        if pat in k["name"]:
            print(f"{k['name'][:100]:100s} vgpr {k['vgpr_count']:3d} agpr {k['agpr_count']:3d} spill v{k['vgpr_spill_count']} s{k['sgpr_spill_count']} "
                  f"scratch {k['private_segment_fixed_size']} B  lds {k['group_segment_fixed_size']}  wg {k['max_flat_workgroup_size']}")
