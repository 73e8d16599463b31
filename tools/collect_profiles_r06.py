#!/usr/bin/env python3
"""Builds profiles/r06_* and the cfg2_f16x2 entry of profiles/r06_traffic.json from what tools/profile_r06.sh left under
gpurun_out/r06.*  (python tools/collect_profiles_r06.py)"""
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
cp("r06.default.json", "r06_bench_default.json")
cp("r06.bf16.json", "r06_bf16_bench.json")
cp("r06.linear.txt", "r06_f_linear_bench.txt")
cp("r06.batch_scaling_bf16.txt", "r06_bf16_shard_timings.txt")
cp("r06.bench_decode.txt", "r06_f_decode_bench.txt")
for a, b in (("f16x2.permuted", "f16x2_permuted_enc"), ("cfg5", "cfg5_f16x2"), ("cfg4", "cfg4_f16x2"), ("ref1024.f16x2", "ref1024_f16x2")):
    cp(f"r06.{a}.json", f"r06_{b}_bench.json")
cp("r06.f16x2.kernel_stats.csv", "r06_f16x2_bench_kernel_stats.csv")
cp("r06.f16x2.under_rocprof.json", "r06_f16x2_bench_under_rocprof.json")
cp("r06.batch_scaling_f16x2.txt", "r06_f16x2_shard_timings.txt")
cp("r06.two_rank.json", "r06_two_rank_rehearsal_one_gpu.json")
with open(os.path.join(dst, "r06_f16x2_hbm_traffic_pmc.txt"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), cfg2, one step, the f16x2 route; KB per launch "
            "(x1024 = bytes; FETCH_SIZE x2 on gfx950 for 16 B/lane streams)\n")
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f.write(open(os.path.join(src, f"r06.f16x2.{c}.txt")).read())
sqfile = "r06_f16x2_sq_counters_pmc.txt"
with open(os.path.join(dst, sqfile), "w") as f:
    f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS "
            "SQ_INSTS_VALU GRBM_GUI_ACTIVE (one pass), cfg2, one step, the SHIPPED f16x2 kernels.  MFMA-pipe utilisation = "
            "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)\n")
    f.write(open(os.path.join(src, "r06.f16x2.SQ.txt")).read())


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


names = {"joint_fwd_gemm": "k_joint_fwd_x2<1>", "dhidden_gemm": "k_dhidden_x2<true", "dw_gemm": "k_dw_x2<4"}
alg = {"joint_fwd_gemm": cells * (4 * H + 4 * V),            # hidden's two planes out, logits out (hidden is never re-read)
       "dhidden_gemm": cells * (4 * V + 4 * V) + 2.5e9,      # logits in, G's two planes out (in place), dEnc/dPred slabs
       "dw_gemm": cells * (4 * V + 4 * H)}                   # G's and hidden's planes in
f_ = counters(os.path.join(src, "r06.f16x2.FETCH_SIZE.txt"), "FETCH_SIZE")
w_ = counters(os.path.join(src, "r06.f16x2.WRITE_SIZE.txt"), "WRITE_SIZE")
busy = counters(os.path.join(dst, sqfile), "SQ_VALU_MFMA_BUSY_CYCLES")
act = counters(os.path.join(dst, sqfile), "GRBM_GUI_ACTIVE")
tpath = os.path.join(dst, "r06_traffic.json")
out = json.load(open(os.path.join(dst, "r05_traffic.json"))) if not os.path.exists(tpath) else json.load(open(tpath))  # (other routes' entries carried over)
out["commit"] = commit
out["cfg2_f16x2"] = {k: {"fetch_raw": pick(f_, n) * 1024, "write": pick(w_, n) * 1024, "algorithmic": alg[k], "commit": commit,
                         "mfma_busy": round(pick(busy, n) / (pick(act, n) / 8 * 1024), 4), "mfma_busy_file": "profiles/" + sqfile}
                     for k, n in names.items()}
json.dump(out, open(tpath, "w"), indent=1)
for s, e in out["cfg2_f16x2"].items():
    print("cfg2_f16x2", s, "traffic %.1f GB (2*%.1f + %.1f) vs algorithmic %.1f GB = %.2fx; write %.1f GB; mfma_busy %.3f" % (
        (2 * e["fetch_raw"] + e["write"]) / 1e9, e["fetch_raw"] / 1e9, e["write"] / 1e9, e["algorithmic"] / 1e9,
        (2 * e["fetch_raw"] + e["write"]) / e["algorithmic"], e["write"] / 1e9, e["mfma_busy"]))
