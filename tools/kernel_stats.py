#!/usr/bin/env python3
"""Per-kernel resource usage of the gfx950 code object inside libinrfit.so (registers, spills, scratch, LDS), and - with
`--isa KERNEL_SUBSTRING` - instruction-class counts of one kernel's disassembly.  Build-container / GPU-box tool; no GPU needed.

    python tools/kernel_stats.py [awesome_amd/csrc/libinrfit.so] [--isa rnvp_fwd_kernelILi3]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def _tool(name):
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else shutil.which(name)


def extract_code_object(lib_path, out_dir):
    objcopy, bundler = _tool("llvm-objcopy"), _tool("clang-offload-bundler")
    if not objcopy or not bundler:
        return None
    fat, co = os.path.join(out_dir, "fatbin"), os.path.join(out_dir, "gfx950.co")
    subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat], check=True)
    subprocess.run([bundler, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return co


def kernel_stats(lib_path):
    readelf = _tool("llvm-readelf")
    if not readelf:
        return None
    with tempfile.TemporaryDirectory() as d:
        co = extract_code_object(lib_path, d)
        if co is None:
            return None
        notes = subprocess.run([readelf, "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = [], None
    keys = ("agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count",
            "vgpr_spill_count")
    for line in notes.splitlines():
        m = re.match(r"\s+- \.(\w+):\s+(.*)", line) or re.match(r"\s+\.(\w+):\s+(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if line.lstrip().startswith("- .") and k in ("agpr_count", "args"):
            cur = {}
            out.append(cur)
        if cur is None:
            continue
        if k == "name":
            cur["name"] = v
        elif k in keys:
            cur[k] = int(v)
    return [k for k in out if "name" in k and "vgpr_count" in k]


def isa_counts(lib_path, needle):
    objdump = _tool("llvm-objdump")
    with tempfile.TemporaryDirectory() as d:
        co = extract_code_object(lib_path, d)
        asm = subprocess.run([objdump, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    res = {}
    blocks = re.split(r"\n(?=[0-9a-f]+ <)", asm)
    for b in blocks:
        head = b.split("\n", 1)[0]
        if needle not in head:
            continue
        name = head.split("<", 1)[1].rsplit(">", 1)[0]
        ins = [ln.split()[0] for ln in b.split("\n")[1:] if ln.startswith("\t") and ln.split()]
        cls = {"total": len(ins), "v_pk_fma": 0, "v_fma/mac": 0, "mfma": 0, "valu_other": 0, "ds": 0, "global/scratch": 0, "scratch": 0,
               "salu/branch": 0, "v_exp/rcp/trans": 0, "v_cndmask": 0, "v_accvgpr": 0}
        for i in ins:
            if i.startswith("v_mfma"):
                cls["mfma"] += 1
            elif i.startswith("v_pk_fma"):
                cls["v_pk_fma"] += 1
            elif i.startswith(("v_fma", "v_mac", "v_fmac")):
                cls["v_fma/mac"] += 1
            elif i.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")):
                cls["v_exp/rcp/trans"] += 1
            elif i.startswith("v_cndmask"):
                cls["v_cndmask"] += 1
            elif i.startswith("v_accvgpr"):
                cls["v_accvgpr"] += 1
            elif i.startswith("v_"):
                cls["valu_other"] += 1
            elif i.startswith("ds_"):
                cls["ds"] += 1
            elif i.startswith("scratch_"):
                cls["scratch"] += 1
                cls["global/scratch"] += 1
            elif i.startswith(("global_", "buffer_", "flat_")):
                cls["global/scratch"] += 1
            elif i.startswith("s_"):
                cls["salu/branch"] += 1
        res[name] = cls
    return res


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    args = [a for a in sys.argv[1:]]
    lib = os.path.join(here, "..", "awesome_amd", "csrc", "libinrfit.so")
    if args and not args[0].startswith("--"):
        lib = args.pop(0)
    if args and args[0] == "--isa":
        for name, c in isa_counts(lib, args[1]).items():
            print(name)
            print("   ", c)
    else:
        for k in sorted(kernel_stats(lib) or [], key=lambda k: k["name"]):
            print(f"{k['name'][:110]:110s} vgpr {k.get('vgpr_count', 0):4d} agpr {k.get('agpr_count', 0):4d} spill {k.get('vgpr_spill_count', 0):3d} "
                  f"scratch {k.get('private_segment_fixed_size', 0):5d} lds {k.get('group_segment_fixed_size', 0):6d}")
