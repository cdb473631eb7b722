"""Turn the rocprofv3 outputs merged into gpurun_out/ (profiles/refresh_profiles.sh) into the committed profiles/r03_* files."""
import csv, glob, json, collections, shutil, os
def newest(pattern):
    """gpurun merges into gpurun_out/ without deleting: take the latest run's file"""
    return max(glob.glob(pattern), key=os.path.getmtime)
R = "/root/repo/gpurun_out"
P = "/root/repo/profiles"
st = newest(R + "/r3_stats/*/*kernel_stats.csv")
shutil.copy(st, P + "/r03_bench_kernel_stats.csv")
ss = glob.glob(R + "/r3_stats_serial/*/*kernel_stats.csv")
if ss: shutil.copy(newest(R + "/r3_stats_serial/*/*kernel_stats.csv"), P + "/r03_bench_serial_kernel_stats.csv")
def pmc(d, name):
    acc = collections.defaultdict(list)
    f = newest(R + "/" + d + "/*/*counter_collection.csv")
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and "vr::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc
fe, wr = pmc("r3_fetch", "FETCH_SIZE"), pmc("r3_write", "WRITE_SIZE")
kern = {}
for k in sorted(set(fe) | set(wr)):
    e = {"launches": len(fe.get(k, wr.get(k)))}
    if k in fe: e["FETCH_SIZE_KiB_per_launch_avg"] = round(sum(fe[k]) / len(fe[k]), 1); e["FETCH_SIZE_KiB_per_launch_max"] = round(max(fe[k]), 1)
    if k in wr: e["WRITE_SIZE_KiB_per_launch_avg"] = round(sum(wr[k]) / len(wr[k]), 1); e["WRITE_SIZE_KiB_per_launch_max"] = round(max(wr[k]), 1)
    kern[k] = e
dk = "vr::k_decode_region"
dec = int((2 * kern[dk]["FETCH_SIZE_KiB_per_launch_avg"] + kern[dk]["WRITE_SIZE_KiB_per_launch_avg"]) * 1024)
# one build(): every encoder kernel's launches of ONE build (the command runs setup builds of the pipelined sets, one
# timed step and the serial passes: all builds are the same work, so per-build = total / number of k_pyramid12 launches)
nbuild = kern["vr::k_pyramid12"]["launches"]
enc = 0
enc_by_kernel = {}
for k, e in kern.items():
    if any(k.startswith(x) for x in ("vr::k_decode", "vr::k_raycast", "vr::k_assemble", "vr::k_disassemble", "vr::k_skip_grid", "vr::k_composite",
                                     "vr::k_measure", "vr::k_query")):
        continue            # (not part of a build)
    b = (2 * e.get("FETCH_SIZE_KiB_per_launch_avg", 0.0) + e.get("WRITE_SIZE_KiB_per_launch_avg", 0.0)) * 1024 * e["launches"] / nbuild
    enc_by_kernel[k] = int(b)
    enc += b
enc = int(enc)
out = {
 "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-render --no-stream",
 "units": "counter values are KiB (rocprofv3 FETCH_SIZE / WRITE_SIZE); bytes = value * 1024",
 "note": "gfx950: WRITE_SIZE is exact for 16-byte-per-lane streaming stores; FETCH_SIZE under-reports wide coalesced reads by 2x and is uncalibrated for other widths (MI355X_MICROARCH.md, HBM section). k_decode_region reads its stream words, the per-4-leaf counts and the depth-(D-3) scalars as 16-byte-per-lane LDS-DMA loads (the 4-byte offsets and 1-byte scalars of the index are the smaller part), so decode_traffic_bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE (the guide's correction); the per-kernel FETCH figures below are raw.",
 "kernels": kern,
 "decode_traffic_bytes_per_launch": dec,
 "encode_traffic_bytes_per_build": enc,
 "encode_traffic_bytes_per_build_by_kernel": dict(sorted(enc_by_kernel.items(), key=lambda kv: -kv[1])),
 "encode_builds_in_run": nbuild,
 "workload": "2048x2048x1920 rm_volume seed 12345, 960 bricks of 256x256x128, tolerance 1, maxEpochs 2",
}
json.dump(out, open(P + "/r03_pmc_hbm_traffic.json", "w"), indent=1)
print("decode traffic", dec, "encode traffic per build", enc, "stats rows", sum(1 for _ in open(P + "/r03_bench_kernel_stats.csv")))
print([l for l in open(R + "/r3_stats.log").read().splitlines() if l.startswith("{\"metric")][-1][:900])
