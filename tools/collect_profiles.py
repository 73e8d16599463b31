#!/usr/bin/env python3
"""Builds profiles/ from what tools/profile_round.sh left under gpurun_out/<tag>.*:
   python tools/collect_profiles.py <tag> [round-prefix, default r01]"""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
B, T, U, H, V = 32, 1000, 200, 512, 1024  # cfg2
cells = B * T * (U + 1)

def cp(a, b):
    shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))

cp(f"{tag}.default.json", f"{rnd}_bench_default.json")
cp(f"{tag}.fp32.under_rocprof.json", f"{rnd}_bench_under_rocprof.json")
cp(f"{tag}.fp32.kernel_stats.csv", f"{rnd}_bench_kernel_stats.csv")
cp(f"{tag}.bf16.default.json", f"{rnd}_bf16_bench_default.json")
cp(f"{tag}.bf16.kernel_stats.csv", f"{rnd}_bf16_bench_kernel_stats.csv")
for dt, out in (("fp32", f"{rnd}_hbm_traffic_pmc.txt"), ("bf16", f"{rnd}_bf16_hbm_traffic_pmc.txt")):
    with open(os.path.join(dst, out), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), cfg2, one step; KB per launch "
                "(x1024 = bytes; FETCH_SIZE x2 on gfx950 for 16 B/lane streams)\n")
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            f.write(open(os.path.join(src, f"{tag}.{dt}.{c}.txt")).read())

def counters(dt, c):
    """kernel-name prefix -> mean counter (KB) of the full-size launches"""
    res, cur = {}, None
    for line in open(os.path.join(src, f"{tag}.{dt}.{c}.txt")):
        if not line.startswith(" "):
            cur = line.strip()
        elif c in line:
            res[cur] = float(line.split()[1])
    return res

def pick(d, key):
    ks = [k for k in d if key in k]
    assert len(ks) == 1, (key, list(d))
    return d[ks[0]]

names = {"cfg2": ("fp32", {"joint_fwd_gemm": "k_joint_fwd_persist<", "dhidden_gemm": "k_dhidden_gen", "dw_gemm": "k_dw<false>"}),
         "cfg2_bf16": ("bf16", {"joint_fwd_gemm": "k_joint_fwd_bf16", "dhidden_gemm": "k_dhidden_bf16", "dw_gemm": "k_dw_bf16"})}
alg = {"cfg2": {"joint_fwd_gemm": cells * (4 * H + 4 * H + 4 * V),  # writes hidden, reads it back, writes logits
                "dhidden_gemm": cells * (4 * V + 4 * H + 4 * V) + 2.5e9,  # logits in, hidden in, G out, dEnc/dPred slabs
                "dw_gemm": cells * (4 * V + 4 * H)},
       "cfg2_bf16": {"joint_fwd_gemm": cells * (2 * H + 2 * V), "dhidden_gemm": cells * (2 * V + 2 * H + 2 * V) + 2.5e9,
                     "dw_gemm": cells * (2 * V + 2 * H)}}
out = {"_note": "HBM-side bytes per launch from rocprofv3 PMC (FETCH_SIZE and WRITE_SIZE collected in separate passes, "
                "KB * 1024), cfg2 on one MI355X.  fetch_raw is the RAW counter: MI355X_MICROARCH.md (HBM section) says "
                "gfx950 FETCH_SIZE reports exactly half the bytes of a 16 B/lane stream; every global load and LDS-DMA of "
                "these kernels is 16 B/lane, so bench.py reports traffic = 2 * fetch_raw + write.  Infinity-Cache hits are "
                "counted as fetches.  algorithmic = bytes the kernel must move once (tools/collect_profiles.py)."}
for key, (dt, ks) in names.items():
    f, w = counters(dt, "FETCH_SIZE"), counters(dt, "WRITE_SIZE")
    out[key] = {}
    for stage, kn in ks.items():
        fr, wr = pick(f, kn), pick(w, kn)  # "k_dw<false>": the contiguous-range walk (full-length batches)
        out[key][stage] = {"fetch_raw": int(fr * 1024), "write": int(wr * 1024), "algorithmic": int(alg[key][stage])}
json.dump(out, open(os.path.join(dst, f"{rnd}_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

# ---- SQ counters of the shipped fp32 kernels + the MFMA utilisation they imply
sq_path = os.path.join(src, f"{tag}.fp32.SQ.txt")
if os.path.exists(sq_path):
    blocks, cur = {}, None
    for line in open(sq_path):
        if not line.startswith(" "):
            cur = line.strip(); blocks[cur] = {}
        else:
            parts = line.split()
            blocks[cur][parts[0]] = float(parts[1])
    with open(os.path.join(dst, f"{rnd}_sq_counters_pmc.txt"), "w") as f:
        f.write("# rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE (one pass, 8 SQ slots), `python3 bench.py --steps 1 --warmup 1`, cfg2 fp32,\n"
                "# means over the full-size launches of each SHIPPED kernel.  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES /\n"
                "# (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): fraction of the chip's matrix-pipe cycles that carried an MFMA.\n")
        for k, c in blocks.items():
            if not any(x in k for x in ("k_joint_fwd_persist", "k_dhidden_gen", "k_dw<false>")):
                continue
            f.write(k + "\n")
            for n, v in sorted(c.items()):
                f.write(f"   {n:32s} {v:.4g}\n")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
                f.write(f"   {'mfma_util':32s} {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.4f}\n")
    print(open(os.path.join(dst, f"{rnd}_sq_counters_pmc.txt")).read())
for extra in ("ref1024.json", "cfg4.json", "ref1024.kernel_stats.csv", "ref1024.bf16.json"):
    if os.path.exists(os.path.join(src, f"{tag}.{extra}")):
        out_name = extra.replace(".bf16", "_bf16").replace(".json", "_bench.json").replace(".kernel_stats.csv", "_kernel_stats.csv")
        cp(f"{tag}.{extra}", f"{rnd}_{out_name}")
