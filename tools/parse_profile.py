"""Turns the rocprofv3 output of tools/profile_gpu.sh (gpurun_out/prof_<tag>/) into the committed
evidence under profiles/:  <tag>_kernel_stats.csv (the --kernel-trace --stats summary),
<tag>_pmc_summary.json (FETCH_SIZE / WRITE_SIZE per launch) and hbm_traffic.json (what bench.py
reports as roofline.traffic).

HBM bytes per launch follow MI355X_MICROARCH.md section "HBM": FETCH_SIZE and WRITE_SIZE are in KiB,
collected in separate --pmc passes; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
coalesced 16 B/lane read stream (which is how Q and A_eq are read), so it is doubled; WRITE_SIZE is
taken as is.   usage: python tools/parse_profile.py <tag> [batch] [workload]
"""
import csv, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
workload = sys.argv[3] if len(sys.argv) > 3 else "W-sat"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))

def per_launch(path, counter):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "fccqp" in r["Kernel_Name"]:
            vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    # a solve of the generic kernel is two launches of one kernel name: the batch, and the (normally
    # empty) list-mode fallback pass; only launches within 10x of the largest count as "the batch"
    big = {k: [x for x in v if x >= 0.1 * max(v)] for k, v in vals.items()}
    return {k: sum(v) / len(v) for k, v in big.items()}, {k: len(v) for k, v in big.items()}

fetch, nf = per_launch(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
write, nw = per_launch(os.path.join(src, "pmc_write", "write_counter_collection.csv"), "WRITE_SIZE")
stats = list(csv.DictReader(open(os.path.join(dst, f"{tag}_kernel_stats.csv"))))
summary = {"tag": tag, "batch": batch, "workload": workload, "kernels": {}}
for k in fetch:
    f_kib, w_kib = fetch[k], write.get(k, 0.0)
    hbm = 2.0 * f_kib * 1024.0 + w_kib * 1024.0
    avg_ns = next((float(s["AverageNs"]) for s in stats if s["Name"] == k), None)
    max_ns = next((float(s["MaxNs"]) for s in stats if s["Name"] == k), None)
    summary["kernels"][k] = {
        "FETCH_SIZE_KiB_per_launch": f_kib, "WRITE_SIZE_KiB_per_launch": w_kib,
        "launches_sampled": [nf[k], nw.get(k, 0)],
        "hbm_bytes_per_launch_corrected": hbm,
        "correction": "2 x FETCH_SIZE (gfx950 counts 128-B requests as 64 B for 16 B/lane streams) + WRITE_SIZE",
        "trace_average_ns": avg_ns, "trace_max_ns": max_ns,
    }
# The rocprofv3 average above runs over EVERY launch of the command, also the untimed ones in front of the timed region
# (parity solve, clock-settling steps, warm-up: bench.py _settle) during which a card coming out of idle is still
# ramping its clocks.  The launches of the timed region itself, from the raw trace: the main kernel's launches in
# start order end with [K timed][min(K, 20) isolated] (bench.py main()).
try:
    bench = json.loads(open(os.path.join(src, "bench_under_trace.json")).read().strip().splitlines()[-1])
    K = int(bench["steps"]); iso = min(K, 20)
    rows = [r for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_trace.csv")))]
    for k in summary["kernels"]:
        mine = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Kernel_Name"] == k)
        big = [e - b for b, e in mine if (e - b) >= 0.1 * max(e2 - b2 for b2, e2 in mine)]
        if len(big) >= K + iso:
            timed = big[-(K + iso):-iso]
            summary["kernels"][k]["trace_timed_region_average_ns"] = sum(timed) / len(timed)
            summary["kernels"][k]["trace_timed_region_launches"] = len(timed)
    summary["bench_under_trace"] = {"kernel_ms_mean": bench["roofline"]["kernel_ms_mean"], "steps": K,
                                    "settle_steps": bench["config"].get("settle_steps")}
except Exception as exc:          # (older traces without the raw csv)
    summary["trace_timed_region_error"] = repr(exc)
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
main = max(summary["kernels"].items(), key=lambda kv: kv[1]["hbm_bytes_per_launch_corrected"])
kernel_id = 2 if "fused" in main[0] else (4 if "coop" in main[0] else 1)
shape = "large" if "large" in tag else "walking"
path = os.path.join(dst, "hbm_traffic.json")
try:
    table = json.load(open(path))
    if "hbm_bytes_per_launch" in table:      # round-1 layout (one entry): start over
        table = {}
except Exception:
    table = {}
table[shape] = {"batch": batch, "workload": workload, "kernel": kernel_id, "tag": tag,
                "hbm_bytes_per_launch": main[1]["hbm_bytes_per_launch_corrected"]}
json.dump(table, open(path, "w"), indent=1)
print(json.dumps(summary, indent=1))
