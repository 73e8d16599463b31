#!/usr/bin/env python3
"""Builds profiles/r04_* from what tools/profile_r04.sh left under gpurun_out/r04.*  (python tools/collect_profiles_r04.py)"""
import json, os, shutil, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def cp(a, b):
    if os.path.exists(os.path.join(src, a)) and os.path.getsize(os.path.join(src, a)) > 0:
        shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
    else:
        print("missing", a)


commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
B, T, U, H, V = 32, 1000, 200, 512, 1024
cells = B * T * (U + 1)
cp("r04.default.json", "r04_bf16x3_bench_default.json")
for a, b in (("fp32", "fp32"), ("bf16", "bf16"), ("bf16x3.permuted", "bf16x3_permuted_enc"), ("cfg5", "cfg5_bf16x3"), ("cfg4", "cfg4_bf16x3"),
             ("ref1024.bf16x3", "ref1024_bf16x3")):
    cp(f"r04.{a}.json", f"r04_{b}_bench.json")
for dt in ("bf16x3", "bf16"):
    cp(f"r04.{dt}.kernel_stats.csv", f"r04_{dt}_bench_kernel_stats.csv")
    cp(f"r04.{dt}.under_rocprof.json", f"r04_{dt}_bench_under_rocprof.json")
cp("r04.batch_scaling_bf16x3.txt", "r04_bf16x3_shard_timings.txt")
cp("r04.bench_decode.txt", "r04_f_decode_bench.txt")
cp("r04.f_decode.kernel_stats.csv", "r04_f_decode_kernel_stats.csv")
cp("r04.mfma_shape.txt", "r04_mfma_shape_probe.txt")
with open(os.path.join(dst, "r04_bf16x3_hbm_traffic_pmc.txt"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), cfg2, one step; KB per launch "
            "(x1024 = bytes; FETCH_SIZE x2 on gfx950 for 16 B/lane streams)\n")
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f.write(open(os.path.join(src, f"r04.bf16x3.{c}.txt")).read())
sqfile = "r04_bf16x3_sq_counters_pmc.txt"
with open(os.path.join(dst, sqfile), "w") as f:
    f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS "
            "SQ_INSTS_VALU GRBM_GUI_ACTIVE (one pass), cfg2, one step, the SHIPPED kernels.  MFMA-pipe utilisation = "
            "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)\n")
    f.write(open(os.path.join(src, "r04.bf16x3.SQ.txt")).read())


def counters(path, c):
    res, cur = {}, None
    for line in open(path):
        if line.startswith("#"):
            continue
        if not line.startswith(" "):
            cur = line.strip()
        elif c in line:
            res[cur] = float(line.split()[1])
    return res


def pick(d, key):
    ks = [k for k in d if key in k]
    assert len(ks) == 1, (key, list(d))
    return d[ks[0]]


names = {"joint_fwd_gemm": "k_joint_fwd_x3", "dhidden_gemm": "k_dhidden_x3<true>", "dw_gemm": "k_dw_x3"}
alg = {"joint_fwd_gemm": cells * (6 * H + 4 * V),            # hidden planes out, logits out (hidden is never re-read)
       "dhidden_gemm": cells * (4 * V + 6 * V) + 2.5e9,      # logits in, G's three planes out, dEnc/dPred slabs
       "dw_gemm": cells * (6 * V + 6 * H)}                   # G's and hidden's planes in
f_, w_ = counters(os.path.join(src, "r04.bf16x3.FETCH_SIZE.txt"), "FETCH_SIZE"), counters(os.path.join(src, "r04.bf16x3.WRITE_SIZE.txt"), "WRITE_SIZE")
busy = counters(os.path.join(dst, sqfile), "SQ_VALU_MFMA_BUSY_CYCLES")
act = counters(os.path.join(dst, sqfile), "GRBM_GUI_ACTIVE")
r03 = json.load(open(os.path.join(dst, "r03_traffic.json")))
out = {"_note": r03["_note"].replace("The fp32 route's kernels are unchanged since round 2: its entry is carried over from profiles/r02_traffic.json.",
                                     "The fp32 and bf16 routes' kernels are unchanged since rounds 2 / 3: their entries are carried over.  mfma_busy = "
                                     "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) of the kernel in the SQ counter pass named by mfma_busy_file."),
       "commit": commit}
out["cfg2_bf16x3"] = {k: {"fetch_raw": pick(f_, n) * 1024, "write": pick(w_, n) * 1024, "algorithmic": alg[k], "commit": commit,
                          "mfma_busy": round(pick(busy, n) / (pick(act, n) / 8 * 1024), 4), "mfma_busy_file": "profiles/" + sqfile}
                      for k, n in names.items()}
for key in ("cfg2_bf16", "cfg2_fp32", "cfg2"):
    out[key] = r03[key]
json.dump(out, open(os.path.join(dst, "r04_traffic.json"), "w"), indent=1)
for s, e in out["cfg2_bf16x3"].items():
    print("cfg2_bf16x3", s, "traffic %.1f GB (2*%.1f + %.1f) vs algorithmic %.1f GB = %.2fx; write %.1f GB; mfma_busy %.3f" % (
        (2 * e["fetch_raw"] + e["write"]) / 1e9, e["fetch_raw"] / 1e9, e["write"] / 1e9, e["algorithmic"] / 1e9,
        (2 * e["fetch_raw"] + e["write"]) / e["algorithmic"], e["write"] / 1e9, e["mfma_busy"]))
