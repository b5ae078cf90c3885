#!/usr/bin/env python3
"""Per-kernel rates of the index build: work counters of the library (scratch/build_phases.py JSON line) over the
rocprofv3 kernel-trace durations of the same run.   usage: condense_build.py <build.log> <kernel_trace.csv> <out.json>"""
import csv, json, sys
log, trace, out = sys.argv[1:4]
model = json.loads([l for l in open(log).read().splitlines() if l.startswith("{")][-1])
k = {}
for r in csv.DictReader(open(trace)):
    for key in ("beam_search_kernel", "pool_prune_kernel", "pool_sort_kernel", "gram_tiles_kernel", "pool_sweep_kernel",
                "backedge_gram_kernel", "backedge_scan_kernel", "backedge_kernel", "bootstrap_kernel", "DeviceRadixSort",
                "make_keys_kernel", "segment_kernel", "seglen_kernel", "set_bulk_kernel"):
        if key in r["kernel"]:
            e = k.setdefault(key, {"calls": 0, "total_ms": 0.0})
            e["calls"] += int(r["calls"])
            e["total_ms"] += float(r["total_ms"])
res = {"model": model, "kernels": k, "rates": {}}
def rate(name, bytes_=None, flop=None):
    if name in k and k[name]["total_ms"] > 0:
        t = k[name]["total_ms"] * 1e-3
        e = res["rates"].setdefault(name, {})
        if bytes_ is not None:
            e["algorithmic_GBps"] = bytes_ / t / 1e9
        if flop is not None:
            e["TFLOPs"] = flop / t / 1e12
            e["frac_of_157TF_f32_mfma_peak"] = flop / t / 157.3e12
rate("beam_search_kernel", bytes_=model["search"]["algorithmic_bytes"])
prune_ms = sum(k[x]["total_ms"] for x in ("pool_prune_kernel", "pool_sweep_kernel", "backedge_kernel", "backedge_gram_kernel") if x in k)
res["rates"]["prune_kernels_row_kernel_pairs"] = {
    "note": "pair + list distances of all prune kernels (pool, back-edge lazy, exact re-checks of the MFMA path) over their summed time; "
            "2 rows per pair, rows mostly L2-resident (just touched by the search)",
    "algorithmic_GBps": model["prune_row_kernel"]["algorithmic_bytes"] / (prune_ms * 1e-3) / 1e9 if prune_ms else None}
# the MFMA flop counter covers both Gram kernels; the tiles kernel's own share = its entries x dim x 2 (the back-edge
# Gram's entries are (list length)^2 per prune, reported by the fused kernel into the same counter)
gk = [x for x in ("gram_tiles_kernel", "backedge_gram_kernel") if x in k]
if len(gk) == 1:
    rate(gk[0], bytes_=model["mfma"]["row_bytes_read"], flop=model["mfma"]["flop"])
elif gk:
    tot_ms = sum(k[x]["total_ms"] for x in gk)
    res["rates"]["gram_kernels_together"] = {"TFLOPs": model["mfma"]["flop"] / (tot_ms * 1e-3) / 1e12,
                                             "frac_of_157TF_f32_mfma_peak": model["mfma"]["flop"] / (tot_ms * 1e-3) / 157.3e12,
                                             "total_ms": tot_ms}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["rates"], indent=1))
