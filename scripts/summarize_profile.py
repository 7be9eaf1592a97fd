#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of scripts/profile.sh (gpurun_out/prof_<tag>/) into the committed summaries:
   profiles/<round>/<key>/kernel_stats_bench_steps3.csv, pmc_per_kernel.csv, dominant_kernel_summary.json
   and the entry <key> of profiles/<round>/pmc_summary.json that bench.py reads (roofline.traffic / mfma_busy).
Usage: scripts/summarize_profile.py gpurun_out/prof_<tag> profiles/r3 <workload key as bench.py's workload_key()>"""
import collections, csv, glob, json, os, shutil, sys

src, root, key = sys.argv[1], sys.argv[2], sys.argv[3]
dst = os.path.join(root, key)
os.makedirs(dst, exist_ok=True)
stats = sorted(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1]
shutil.copy(stats, os.path.join(dst, "kernel_stats_bench_steps3.csv"))
rows = list(csv.DictReader(open(stats)))
total_ns = sum(float(r["TotalDurationNs"]) for r in rows)

agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
per_disp = []                       # dispatches of the pass that carries MFMA-busy AND GRBM_GUI_ACTIVE: (kernel, duration ns, counters)
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    fs = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        continue
    f = fs[-1]                      # (a re-profiled tag keeps older passes next to the new one: the newest pass only)
    byd = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k, c = r["Kernel_Name"], r["Counter_Name"]
        agg[k][c] += float(r["Counter_Value"])
        disp[k][c].add(r["Dispatch_Id"])
        byd[r["Dispatch_Id"]][c] = float(r["Counter_Value"])
        byd[r["Dispatch_Id"]]["_k"] = k
    if any("GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v for v in byd.values()):
        tr = f.replace("counter_collection.csv", "kernel_trace.csv")
        dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr))} if os.path.exists(tr) else {}
        per_disp = [(v["_k"], dur.get(i, 0), v) for i, v in byd.items()]
with open(os.path.join(dst, "pmc_per_kernel.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "dispatches", "sum", "per_dispatch"])
    for k in sorted(agg):
        for c in sorted(agg[k]):
            n = len(disp[k][c])
            w.writerow([k, c, n, f"{agg[k][c]:.6g}", f"{agg[k][c] / max(n, 1):.6g}"])


def family(pred):
    sel = [r for r in rows if pred(r["Name"])]
    calls = sum(int(r["Calls"]) for r in sel)
    ns = sum(float(r["TotalDurationNs"]) for r in sel)
    cnt = collections.defaultdict(float)
    nd = 0
    for k in agg:
        if pred(k):
            for c, v in agg[k].items():
                cnt[c] += v
            nd += len(disp[k].get("FETCH_SIZE", ()))
    out = {"launches": calls, "avg_launch_us_rocprof": ns / max(calls, 1) / 1e3, "share_of_kernel_time": ns / total_ns, "pmc_dispatches": nd}
    if nd:
        out["hbm_bytes_per_launch_pmc"] = (2 * cnt["FETCH_SIZE"] + cnt["WRITE_SIZE"]) * 1024 / nd
    if cnt.get("GRBM_GUI_ACTIVE"):
        out["mfma_busy_fraction"] = cnt["SQ_VALU_MFMA_BUSY_CYCLES"] / (cnt["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
    if cnt.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_fraction"] = cnt["SQ_LDS_BANK_CONFLICT"] / cnt["SQ_LDS_IDX_ACTIVE"]
    if cnt.get("TCC_HIT_sum"):
        out["l2_hit_rate"] = cnt["TCC_HIT_sum"] / (cnt["TCC_HIT_sum"] + cnt["TCC_MISS_sum"])
    # Dispatches of >= 0.3 ms only, MFMA-busy and clock from ONE pass (GRBM_GUI_ACTIVE / 8 / time reads high on shorter dispatches,
    # MI355X_MICROARCH.md 'DVFS give-back'): busy = MFMA-busy cycles / (SIMD cycles), clock = GUI cycles / time, and the fraction of
    # the 2.5 PFLOP/s peak these dispatches EXECUTE (padding included; 1024 FLOP per MFMA-busy cycle for both bf16 shapes) -- which
    # must equal busy x clock / 2.4 GHz (peak = 1024 SIMDs x 1024 FLOP x 2.4 GHz = 2.517 PF vs the nominal 2.5: 0.7 %).
    long = [(d, v) for k, d, v in per_disp if pred(k) and d >= 300000 and "GRBM_GUI_ACTIVE" in v]
    if long:
        busy_c = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for _, v in long)
        gui = sum(v["GRBM_GUI_ACTIVE"] for _, v in long) / 8.0
        t = sum(d for d, _ in long) * 1e-9
        out["long_dispatches"] = len(long)
        out["long_share_of_family_time"] = sum(d for d, _ in long) / max(1.0, sum(d for k, d, v in per_disp if pred(k)))
        out["mfma_busy_long"] = busy_c / (gui * 256 * 4)
        out["clock_ghz_long"] = gui / t / 1e9
        out["frac_executed_long"] = busy_c * 1024 / t / 2.5e15
        recon = out["mfma_busy_long"] * out["clock_ghz_long"] / 2.4
        assert abs(recon - out["frac_executed_long"]) <= 0.05 * out["frac_executed_long"], (recon, out["frac_executed_long"])
        out["busy_x_clock_over_2p4"] = recon
    return out


cmd = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"


def calls_of(*needles):
    return sum(int(r["Calls"]) for r in rows if any(n in r["Name"] for n in needles))


# what the profiled command ran, from its own launch counts: one optimizer launch per train step, one metadata-MLP forward per
# network forward (train or eval) -- warm-up, capture, timed regions, the read-back pass, the event pass and the latency forwards
n_steps = calls_of("adamw_pack_kernel") or calls_of("mse_kernel")
n_fwd = calls_of("meta_mlp_fwd_kernel")
ran = f"{n_steps} train steps + {max(0, n_fwd - n_steps)} eval-mode forwards, counted from the launches of this profile"
summary = {
    "command": f"rocprofv3 --kernel-trace [--stats | --pmc <set>] --output-format csv -- {cmd}   ({ran}; scripts/profile.sh, scripts/summarize_profile.py)",
    "train_steps_in_profile": n_steps, "eval_forwards_in_profile": max(0, n_fwd - n_steps),
    "correction": "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE tallies 128-B requests as 64 B for wide coalesced reads incl. LDS-DMA; WRITE_SIZE is exact for 16-B stores (MI355X_MICROARCH.md, HBM section)",
    "mfma_busy_fraction": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs)",
    "conv3x3_bf16_kernel (dominant: forward + data gradient)": family(lambda n: "conv3x3_bf16_kernel" in n),
    "wgrad kernels (wgrad16_kernel + wgrad_bf16_kernel)": family(lambda n: "wgrad16_kernel" in n or "wgrad_bf16_kernel" in n),
}
json.dump(summary, open(os.path.join(dst, "dominant_kernel_summary.json"), "w"), indent=1)
dom = summary["conv3x3_bf16_kernel (dominant: forward + data gradient)"]
pj = os.path.join(root, "pmc_summary.json")
allk = json.load(open(pj)) if os.path.exists(pj) else {}
import hashlib
libp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "metadata-augmented-unet-for-lst-ndvi_amd", "libmau_hip.so")
lib_sha = hashlib.sha256(open(libp, "rb").read()).hexdigest()       # bench.py quotes these counters only for the SAME binary
allk[key] = {"lib_sha256": lib_sha, "hbm_bytes_per_launch": dom.get("hbm_bytes_per_launch_pmc"), "mfma_busy": dom.get("mfma_busy_fraction"),
             "avg_launch_us_rocprof": dom["avg_launch_us_rocprof"], "launches": dom["launches"], "l2_hit_rate": dom.get("l2_hit_rate"),
             "lds_bank_conflict_fraction": dom.get("lds_bank_conflict_fraction"), "command": cmd,
             "mfma_busy_long": dom.get("mfma_busy_long"), "clock_ghz_long": dom.get("clock_ghz_long"), "frac_long": dom.get("frac_executed_long"),
             "long_dispatches": dom.get("long_dispatches"), "long_share_of_family_time": dom.get("long_share_of_family_time"),
             "wgrad": summary["wgrad kernels (wgrad16_kernel + wgrad_bf16_kernel)"]}
json.dump(allk, open(pj, "w"), indent=1)
print(json.dumps(summary, indent=1))
top = sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]
for r in top:
    print(f"{r['Name'][:80]:80s} {int(r['Calls']):5d} {float(r['TotalDurationNs']) / total_ns * 100:5.1f}%")
