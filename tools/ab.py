#!/usr/bin/env python3
"""A/B of library builds on one box: `python tools/ab.py main nosgb ...` runs tools/kbench.py once per build (INRFIT_LIB), each in
its own process; "main" = the shipped awesome_amd/csrc/libinrfit.so, anything else = variants/libinrfit_<name>.so."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
extra = [a for a in sys.argv[1:] if a.startswith("--")]
names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["main"]
for rep in range(2):
    for n in names:
        env = dict(os.environ)
        if n != "main":
            env["INRFIT_LIB"] = os.path.join(ROOT, "variants", f"libinrfit_{n}.so")
            env["INRFIT_ABI_ANY"] = "1"
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--sizes", "256", "--images", "1", "64", "--fit-steps", "1000"] + extra,
                             env=env, capture_output=True, text=True)
        for line in out.stdout.splitlines():
            if line.startswith("size"):
                print(f"[{n:10s}] {line}", flush=True)
        if out.returncode:
            print(out.stderr[-800:])
