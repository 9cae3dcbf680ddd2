"""Build libinrfit.so (hipcc, gfx950) in-tree.  `python -m awesome_amd.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "inrfit.hip")
OUT = os.path.join(HERE, "csrc", "libinrfit.so")
INCLUDE = os.path.join(ROOT, "include")


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    deps = [SRC, os.path.join(INCLUDE, "inrfit.h")] + [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".h")]
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile the HIP C-ABI library for gfx950.  Cross-compiles without a GPU."""
    if not force and not needs_build():
        return OUT
    # -fno-slp-vectorize: the SLP pass packs the leftover-unit FMAs into v_pk_fma_f32 and pays two v_mov per pack for it;
    # -amdgpu-mfma-vgpr-form: MFMA results in VGPRs where the allocation allows, instead of AGPRs read back with v_accvgpr_read.
    # Together 14 % fewer VALU instructions in the step kernels' chunk loop (every one of them costs MFMA issue slots),
    # no spills left in the L = 2 kernels: step kernel -1 %.
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form",
           "-shared", "-fPIC", f"-I{INCLUDE}", SRC, "-o", OUT]
    if verbose:
        print("[awesome_amd.build]", " ".join(cmd), flush=True)
    if subprocess.run(cmd).returncode != 0:
        # the two code-generation switches are tuning only (and -mllvm options are not a stable interface): build without them
        plain = [c for c in cmd if c not in ("-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form")]
        if verbose:
            print("[awesome_amd.build] retrying without the code-generation switches:", " ".join(plain), flush=True)
        subprocess.run(plain, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
