#!/usr/bin/env python3
"""Turn the on-box summaries of tools/profile_r02.sh / profile_round.sh (gpurun_out/rNN_prof*/) into the committed profiles/rNN_<tag>_* files.

    python tools/make_profile_summary.py <tag> [<source dir under gpurun_out> [<round>]]      e.g.  ... e r03_prof_e 3"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r02_prof")
DST = os.path.join(ROOT, "profiles")
SES = 32   # SQ counters come per shader engine: 32 instances per dispatch (8 CUs = 32 SIMDs each)


def pmc(path):
    out, cur = {}, None
    for line in open(path):
        if line.startswith("#"):
            continue
        if not line.startswith(" "):
            cur = line.strip()
            out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)", line)
            out[cur][m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return out


def find(d, frag):
    return next(v for k, v in d.items() if frag in k)


def stats(path):
    return {r["Name"]: r for r in csv.DictReader(open(path))}


def main():
    global SRC
    tag = sys.argv[1] if len(sys.argv) > 1 else "a"
    if len(sys.argv) > 2:
        SRC = os.path.join(ROOT, "gpurun_out", sys.argv[2])
    rnd = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    R = f"r{rnd:02d}"
    if os.path.exists(os.path.join(SRC, "bench_line.json")):
        shutil.copy(os.path.join(SRC, "bench_line.json"), os.path.join(DST, f"{R}_{tag}_bench_line.json"))
    for src, name, cmd in (("bench", "bench", "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --strong-images 0 --no-variants --throughput-images 0"),
                           ("pcn", "pcn_fit", "python3 tools/kbench_pcn.py --steps 200"), ("cdn", "cdn_fit", "python3 tools/kbench_cdn.py"),
                           ("joint", "joint_step", "python3 tools/kbench_joint.py")):
        if not os.path.exists(os.path.join(SRC, f"{src}_kernel_stats.csv")):
            continue
        with open(os.path.join(DST, f"{R}_{tag}_kernel_stats_{name}.csv"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}   (table from the rocpd file by tools/rocpd_stats.py)\n")
            f.write(open(os.path.join(SRC, f"{src}_kernel_stats.csv")).read())
    # ---- step kernel: traffic + SQ counters (skipped when this collection did not run the bench PMC passes)
    rec = {"derived": {}}
    if os.path.exists(os.path.join(SRC, "bench_pmc_sq.txt")):
        rec = step_kernel_summary(tag, rnd, R)
    if not os.path.exists(os.path.join(SRC, "pcn_pmc_sq.txt")):
        print(json.dumps(rec["derived"], indent=1))
        return
    rnvp_summary(tag, rnd, R, rec)


def step_kernel_summary(tag, rnd, R):
    fs, ws, sq = (pmc(os.path.join(SRC, f"bench_pmc_{c}.txt")) for c in ("FETCH_SIZE", "WRITE_SIZE", "sq"))
    k = "icnn_step_kernel<130, 2, true"
    f_kb, w_kb = find(fs, k)["FETCH_SIZE"][1], find(ws, k)["WRITE_SIZE"][1]
    s = {c: v[1] for c, v in find(sq, k).items()}
    waves_per_se = 1024 / SES
    dur = s["SQ_BUSY_CYCLES"]
    st = stats(os.path.join(SRC, "bench_kernel_stats.csv"))
    avg_ns = float(next(v for kk, v in st.items() if k in kk)["AverageNs"])
    life = 4 * s["SQ_WAVE_CYCLES"] / waves_per_se
    rec = {
        "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_* (three separate passes, no trace domain besides the counters) -- "
                   "python3 bench.py --steps 1 --warmup 0 --epochs 50 --kernel-iters 20 --no-cpu-baseline --throughput-images 0 --no-variants",
        "round": rnd, "kernel": "icnn_step_kernel<130,2,train,relu>", "workload": "1 x 256x256, 256 workgroups x 4 waves",
        "kernel_avg_us_rocprofv3_kernel_trace": round(avg_ns / 1e3, 2),
        "FETCH_SIZE_KB_mean": f_kb, "WRITE_SIZE_KB_mean": w_kb,
        "correction": "MI355X_MICROARCH.md HBM: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads -> x2; WRITE_SIZE exact",
        "traffic_bytes_per_launch": int(round((2 * f_kb + w_kb) * 1024)),
        "algorithmic_bytes_per_launch": {"targets": 262144, "parameter_image_first_touch_per_XCD": 8 * 78848, "gradient_slabs": 256 * 17816 * 4},
        "update_kernel": {"FETCH_SIZE_KB_mean": find(fs, "icnn_update_kernel")["FETCH_SIZE"][1],
                          "WRITE_SIZE_KB_mean": find(ws, "icnn_update_kernel")["WRITE_SIZE"][1]},
        "sq_counters_per_shader_engine_mean": s,
        "derived": {
            "mfma_per_wave": s["SQ_INSTS_MFMA"] / waves_per_se,
            "non_mfma_valu_per_wave": (s["SQ_INSTS_VALU"] - s["SQ_INSTS_MFMA"]) / waves_per_se,
            "wave_lifetime_cycles": life,
            "matrix_pipe_busy_frac_of_kernel_duration": round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / 32 / dur, 4),
            "matrix_pipe_busy_frac_of_wave_lifetime": round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / 32 / life, 4),
            "parked_at_waitcnt_or_barrier_frac": round(s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"], 4),
            "issue_stall_frac": round(s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"], 4),
            "note": "SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles, SQ_BUSY_CYCLES and SQ_VALU_MFMA_BUSY_CYCLES cycles; one wave per SIMD; "
                    "issue stalls = waiting for the matrix pipe, expected when MFMA-bound",
        },
    }
    json.dump(rec, open(os.path.join(DST, f"{R}_{tag}_pmc_step_kernel.json"), "w"), indent=1)
    return rec


def rnvp_summary(tag, rnd, R, rec):
    # ---- RealNVP kernels at configs[3]
    fs, ws, sq = (pmc(os.path.join(SRC, f"pcn_pmc_{c}.txt")) for c in ("FETCH_SIZE", "WRITE_SIZE", "sq"))
    st = stats(os.path.join(SRC, "pcn_kernel_stats.csv"))
    N, F, C, HID = 128 * 128 * 16, 18, 3, 32
    fwd_flop = F * HID * 2 * C * 2                       # per point: F flows x hid units x 2 nets x (NIN + NOUT = C) fma
    alg = {   # algorithmic (flop, bytes) per point and launch
        "rnvp_fwd_kernel<3": (fwd_flop, 4 * (C + C + F * C)),                                       # coords in, xd out, zs (state before every flow) out
        "rnvp_bwd_points_kernel<3": (2 * fwd_flop + F * HID * 2 * 2 * 2, 4 * (C + F * C + F * 3)),   # forward again + input Jacobian; dxd, zs in, ps out
        "rnvp_bwd_units_kernel<3": (F * HID * 2 * (1 + 1 + 2) * 2, 4 * (F * 3 + F * 1.5)),           # moment sums; ps + the active zs channels in
    }
    out = {"command": "rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_* (separate passes) -- python3 tools/kbench_pcn.py --case xyt",
           "round": rnd, "workload": "BASELINE configs[3]: PathConnectedNet C=3, 18 flows x 32 hidden units, 128x128x16 = 262144 points, one optimizer step",
           "peaks": {"fp32_vector_TFLOPs": 157.3, "hbm_TBs": 8.0}, "kernels": {}}
    for kname, (flop_pt, bytes_pt) in alg.items():
        ns = float(next(v for kk, v in st.items() if kname in kk)["AverageNs"])
        s = {c: v[1] for c, v in find(sq, kname).items()}
        fb, wb = 2 * find(fs, kname)["FETCH_SIZE"][1] * 1024, find(ws, kname)["WRITE_SIZE"][1] * 1024
        out["kernels"][kname[:-2] + "<3>"] = {
            "avg_us": round(ns / 1e3, 1),
            "algorithmic_flop_per_point": flop_pt, "achieved_TFLOPs": round(flop_pt * N / ns / 1e3, 2),
            "frac_of_fp32_vector_peak": round(flop_pt * N / ns / 1e3 / 157.3, 4),
            "algorithmic_bytes_per_point": bytes_pt, "measured_hbm_bytes_per_launch": int(fb + wb),
            "achieved_TBs": round((fb + wb) / ns / 1e3, 3), "frac_of_hbm_peak": round((fb + wb) / ns / 1e3 / 8.0, 4),
            "valu_busy_frac": round(4 * s["SQ_INSTS_VALU"] / 32 / s["SQ_BUSY_CYCLES"], 4),
            "valu_wave_instructions_per_simd": round(s["SQ_INSTS_VALU"] / 32, 1), "lds_wave_instructions_per_simd": round(s["SQ_INSTS_LDS"] / 32, 1),
            "valu_instructions_per_wave_of_64_points_and_flow": round(s["SQ_INSTS_VALU"] * 32 / (N / 64.0) / F, 1),
            "lds_instructions_per_wave_of_64_points_and_flow": round(s["SQ_INSTS_LDS"] * 32 / (N / 64.0) / F, 1),
            "waves_per_simd_resident_mean": round(4 * s["SQ_WAVE_CYCLES"] / 32 / s["SQ_BUSY_CYCLES"], 2),
            "bound": ("VALU issue (a wave64 VALU instruction occupies its SIMD for 4 cycles); neither the flop nor the HBM roof is near" if rnd < 3 else
                      "VALU issue at 3-4 waves per SIMD (0.6-0.7 busy) next to LDS / memory latency; neither the flop nor the HBM roof is near "
                      "(profiles/NOTES.md, round 3: the scalar data path and Q points per lane were slower; fewer VALU instructions per unit - "
                      "relu / step through the clamp modifier - made the backward kernels faster)"),
        }
    json.dump(out, open(os.path.join(DST, f"{R}_{tag}_pmc_rnvp_kernels.json"), "w"), indent=1)
    print(json.dumps(rec["derived"], indent=1))
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
