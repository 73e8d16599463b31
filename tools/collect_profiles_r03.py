#!/usr/bin/env python3
"""Builds profiles/r03_* from what tools/profile_r03.sh left under gpurun_out/r03.*  (python tools/collect_profiles_r03.py)"""
import json, os, shutil, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
def cp(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
    else:
        print("missing", a)
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
B, T, U, H, V = 32, 1000, 200, 512, 1024
cells = B * T * (U + 1)
cp("r03.default.json", "r03_bench_default.json")
for a, b in (("fp32", "fp32"), ("bf16", "bf16"), ("bf16x3.permuted", "bf16x3_permuted_enc"), ("fp32.permuted", "fp32_permuted_enc"),
             ("cfg5", "cfg5_bf16x3")):
    cp(f"r03.{a}.json", f"r03_{b}_bench.json")
for dt in ("bf16x3", "fp32", "bf16"):
    cp(f"r03.ref1024.{dt}.json", f"r03_ref1024_{dt}_bench.json")
    cp(f"r03.ref1024.{dt}.permuted.json", f"r03_ref1024_{dt}_permuted_enc_bench.json")
    cp(f"r03.{dt}.kernel_stats.csv", f"r03_{dt}_bench_kernel_stats.csv")
    cp(f"r03.{dt}.under_rocprof.json", f"r03_{dt}_bench_under_rocprof.json")
for dt in ("bf16x3", "bf16"):
    with open(os.path.join(dst, f"r03_{dt}_hbm_traffic_pmc.txt"), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), cfg2, one step; KB per launch "
                "(x1024 = bytes; FETCH_SIZE x2 on gfx950 for 16 B/lane streams)\n")
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            f.write(open(os.path.join(src, f"r03.{dt}.{c}.txt")).read())
    with open(os.path.join(dst, f"r03_{dt}_sq_counters_pmc.txt"), "w") as f:
        f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS "
                "SQ_INSTS_VALU GRBM_GUI_ACTIVE (one pass), cfg2, one step, the SHIPPED kernels.  MFMA-pipe utilisation = "
                "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)\n")
        f.write(open(os.path.join(src, f"r03.{dt}.SQ.txt")).read())
def counters(dt, c):
    res, cur = {}, None
    for line in open(os.path.join(src, f"r03.{dt}.{c}.txt")):
        if not line.startswith(" "):
            cur = line.strip()
        elif c in line:
            res[cur] = float(line.split()[1])
    return res
def pick(d, key):
    ks = [k for k in d if key in k]
    assert len(ks) == 1, (key, list(d))
    return d[ks[0]]
names = {"cfg2_bf16x3": ("bf16x3", {"joint_fwd_gemm": "k_joint_fwd_x3", "dhidden_gemm": "k_dhidden_x3<true>", "dw_gemm": "k_dw_x3"}),
         "cfg2_bf16": ("bf16", {"joint_fwd_gemm": "k_joint_fwd_bf16", "dhidden_gemm": "k_dhidden_bf16<true>", "dw_gemm": "k_dw_bf16"})}
alg = {"cfg2_bf16x3": {"joint_fwd_gemm": cells * (6 * H + 4 * V),            # hidden planes out, logits out (hidden is never re-read)
                       "dhidden_gemm": cells * (4 * V + 6 * V) + 2.5e9,      # logits in, G's three planes out, dEnc/dPred slabs
                       "dw_gemm": cells * (6 * V + 6 * H)},                  # G's and hidden's planes in
       "cfg2_bf16": {"joint_fwd_gemm": cells * (2 * H + 2 * V), "dhidden_gemm": cells * (2 * V + 2 * H + 2 * V) + 2.5e9,
                     "dw_gemm": cells * (2 * V + 2 * H)}}
out = {"_note": "HBM-side bytes per launch from rocprofv3 PMC (FETCH_SIZE and WRITE_SIZE collected in separate passes, KB * 1024), cfg2 on one "
                "MI355X.  fetch_raw is the RAW counter: MI355X_MICROARCH.md (HBM section) says gfx950 FETCH_SIZE reports exactly half the bytes "
                "of a 16 B/lane stream; every global load and LDS-DMA of these kernels is 16 B/lane, so bench.py reports traffic = 2 * fetch_raw "
                "+ write.  Infinity-Cache hits are counted as fetches.  algorithmic = bytes the kernel must move once.  The fp32 route's kernels "
                "are unchanged since round 2: its entry is carried over from profiles/r02_traffic.json.",
       "commit": commit}
for key, (dt, ks) in names.items():
    f, w = counters(dt, "FETCH_SIZE"), counters(dt, "WRITE_SIZE")
    out[key] = {k: {"fetch_raw": pick(f, n) * 1024, "write": pick(w, n) * 1024, "algorithmic": alg[key][k], "commit": commit} for k, n in ks.items()}
r02 = json.load(open(os.path.join(dst, "r02_traffic.json")))
out["cfg2_fp32"] = out["cfg2"] = {k: dict(v, commit="e73e6fb (round 2: kernels unchanged)") for k, v in r02["cfg2"].items()}
json.dump(out, open(os.path.join(dst, "r03_traffic.json"), "w"), indent=1)
cp("r03.bench_predictor.txt", "r03_f_predictor_bench.txt")
cp("r03.bench_optim.json", "r03_f_optim_bench.json")
cp("r03.bench_decode.txt", "r03_f_decode_bench.txt")
for t in ("predictor", "optim", "decode"):
    cp(f"r03.f_{t}.kernel_stats.csv", f"r03_f_{t}_kernel_stats.csv")
cp("exp_bf16_traffic.json", "r03_bf16_ablation_loads_stores_only.json")
for k, v in out.items():
    if isinstance(v, dict):
        for s, e in v.items():
            print(k, s, "traffic %.1f GB (2*%.1f + %.1f) vs algorithmic %.1f GB = %.2fx" % ((2 * e["fetch_raw"] + e["write"]) / 1e9, e["fetch_raw"] / 1e9, e["write"] / 1e9, e["algorithmic"] / 1e9, (2 * e["fetch_raw"] + e["write"]) / e["algorithmic"]))
