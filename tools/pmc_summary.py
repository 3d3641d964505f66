"""HBM traffic per contraction launch from two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
Units and the gfx950 correction follow MI355X_MICROARCH.md: both counters are in KB (x1024), FETCH_SIZE tallies 128-byte
requests as 64 bytes (x2).  usage: pmc_summary.py <dir with pmc_fetch/ and pmc_write/> <out.json>"""
import csv, glob, json, sys

src, out = sys.argv[1], sys.argv[2]


def collect(sub, counter):
    f = glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        if "conv_gemm_f32" in k or "wn_layer" in k:
            tot += float(r["Counter_Value"])
            n += 1
        elif "winograd_" in k:  # transforms of a Winograd-form conv: same logical launch as its contraction
            tot += float(r["Counter_Value"])
    return tot, n


fetch, nf = collect("pmc_fetch", "FETCH_SIZE")
write, nw = collect("pmc_write", "WRITE_SIZE")
res = {
    "kernels": "conv_gemm_f32 + wn_layer_kernel (+ Winograd transforms)",
    "launches_sampled": nf,
    "fetch_bytes_per_launch_corrected_x2": int(fetch * 1024 * 2 / nf),
    "write_bytes_per_launch": int(write * 1024 / nw),
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1` (KB units x1024); "
            "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B; counts L2-miss traffic incl. Infinity-Cache hits)",
}
res["hbm_bytes_per_launch"] = res["fetch_bytes_per_launch_corrected_x2"] + res["write_bytes_per_launch"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
