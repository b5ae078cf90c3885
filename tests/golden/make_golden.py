#!/usr/bin/env python3
"""Extract the reference's own golden vectors for the hot path into small fixtures.

Runs ONLY in the build container (needs /root/reference).  The GPU box has no
reference tree, so tests read the fixtures committed next to this script.

Sources (all plain text in the reference tree):
  * diskann/test/generated/graph/test/cases/grid_search/*.json   (18 beam-search cases)
  * diskann/test/generated/graph/test/cases/grid_insert/**.json  (insert+search cases)
  * diskann-wide/test_data/float16_conversion.txt                (f16 -> f32, exhaustive)
Only test *data* is extracted (queries, expected ids/distances/counters); no source code.
"""
import glob
import json
import os
import struct

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def grid_search():
    out = []
    for path in sorted(glob.glob(f"{REF}/diskann/test/generated/graph/test/cases/grid_search/*.json")):
        doc = json.load(open(path))
        for case in doc["payload"]:
            out.append({
                "source": os.path.relpath(path, REF),
                "grid_dims": case["grid_dims"],
                "grid_size": case["grid_size"],
                "beam_width": case["beam_width"],
                "l_value": 10,  # Knn::new(10, Some(beam_width)), grid_search.rs:131
                "k": 10,
                "query": case["query"],
                "results": case["results"],
                "comparisons": case["comparisons"],
                "hops": case["hops"],
                "num_results": case["num_results"],
            })
    json.dump(out, open(f"{HERE}/grid_search.json", "w"), indent=0, separators=(",", ":"))
    print("grid_search cases:", len(out))


def grid_insert():
    out = []
    for path in sorted(glob.glob(f"{REF}/diskann/test/generated/graph/test/cases/grid_insert/**/*.json",
                                 recursive=True)):
        doc = json.load(open(path))
        out.append({"source": os.path.relpath(path, REF), "test": doc.get("test"), "payload": doc["payload"]})
    json.dump(out, open(f"{HERE}/grid_insert.json", "w"), separators=(",", ":"))
    print("grid_insert files:", len(out))


def range_search():
    out = []
    for path in sorted(glob.glob(f"{REF}/diskann/test/generated/graph/test/cases/range_search/*.json")):
        doc = json.load(open(path))
        c = doc["payload"]
        # parameters that are not stored in the baseline come from the test source
        # (diskann/src/graph/test/cases/range_search.rs:157-470): starting_l is stored; max_returned
        # only in the two max_results tests (4 and 5)
        name = os.path.basename(path)[:-5]
        max_returned = {"max_results_respected_means_no_second_round": 4,
                        "max_results_respected_and_second_round_triggered": 5}.get(name, 0)
        out.append({"source": os.path.relpath(path, REF), "name": name, "grid_dims": c["grid_dims"],
                    "grid_size": c["grid_size"], "query": c["query"], "radius": c["radius"],
                    "inner_radius": c["inner_radius"], "starting_l": c["starting_l"], "max_returned": max_returned,
                    "results": c["results"], "comparisons": c["comparisons"], "hops": c["hops"],
                    "result_count": c["result_count"], "second_round": c["range_search_second_round"]})
    json.dump(out, open(f"{HERE}/range_search.json", "w"), separators=(",", ":"))
    print("range_search cases:", len(out))


def filtered():
    """inline / multihop / filtered_range goldens.  Parameters that the JSON does not store come from the
    test sources (diskann/src/graph/test/cases/{inline,multihop,filtered_range_search}.rs)."""
    base = f"{REF}/diskann/test/generated/graph/test/cases"
    out = {"inline": [], "multihop": [], "filtered_range": []}
    # inline.rs Setup1D (:160-243): filter sets and AdaptiveL parameters per scenario
    setups = {"no_scaling": (list(range(40, 100)), (5, 16.0)), "linear": ([43, 44, 92, 95], (10, 16.0)),
              "logarithmic": ([43, 95], (20, 16.0)), "max": ([10, 20, 30, 50], (5, 16.0))}
    for path in sorted(glob.glob(f"{base}/inline/*.json")):
        name = os.path.basename(path)[:-5]
        c = json.load(open(path))["payload"]
        case = {"source": os.path.relpath(path, REF), "name": name, "query": c["query"], "k": c["k"], "l": c["l"],
                "comparisons": c["comparisons"], "hops": c["hops"], "result_count": c["result_count"]}
        if "results" in c:
            case["result_ids"] = [r[0] for r in c["results"]]
            case["result_distances"] = [r[1] for r in c["results"]]
        else:
            case["result_ids"], case["result_distances"] = c["result_ids"], c["result_distances"]
        if name.startswith("inline_adaptive_l_") or name.startswith("inline_fixed_"):
            kind, scen = name.split("_")[1], name.split("_", 2 if name.startswith("inline_fixed") else 3)[-1]
            case.update(graph="grid1d_100", filter=setups[scen][0],
                        adaptive=list(setups[scen][1]) if kind == "adaptive" else None)
        elif "three_level" in name or "final_level" in name:
            # inline.rs:56-110 (three-level tree, labels: ids 7..14 match), :382-528
            case.update(graph="three_level", filter=list(range(7, 15)),
                        adaptive=[1, 16.0] if "three_level_adaptive" in name else None)
        else:  # inline.rs:598-670: 1-D hand-built graph, EvenFilter (start id 10 is even)
            case.update(graph="hand_1d", filter="even", adaptive=None)
        out["inline"].append(case)
    for path in sorted(glob.glob(f"{base}/multihop/*.json")):
        name = os.path.basename(path)[:-5]
        c = json.load(open(path))["payload"]
        out["multihop"].append({"source": os.path.relpath(path, REF), "name": name, "query": c["query"], "k": c["k"],
                                "l": c["l"], "grid_size": c["grid_size"], "results": c["results"],
                                "comparisons": c["comparisons"], "hops": c["hops"],
                                "graph": "hand_1d" if c["grid_size"] == 0 else "grid3d", "filter": "even"})
    # filtered_range_search.rs: filters and max_returned per test (:88-446)
    fr = {"basic_range_search": ("all", 0), "inner_radius_filtering": ("all", 0), "two_round_search": ("all", 0),
          "max_results_respected_means_no_second_round": ("all", 4),
          "max_results_respected_and_second_round_triggered": ("all", 200),
          "divisible_by_four_filter_second_round_triggered": ("div4", 0),
          "divisible_by_four_filter_no_second_round_from_l_search": ("div4", 0)}
    for path in sorted(glob.glob(f"{base}/filtered_range_search/*.json")):
        name = os.path.basename(path)[:-5]
        c = json.load(open(path))["payload"]
        out["filtered_range"].append({
            "source": os.path.relpath(path, REF), "name": name, "grid_dims": c["grid_dims"],
            "grid_size": c["grid_size"], "query": c["query"], "radius": c["radius"],
            "inner_radius": c["inner_radius"], "starting_l": c["starting_l"], "filter": fr[name][0],
            "max_returned": fr[name][1], "results": c["results"], "comparisons": c["comparisons"], "hops": c["hops"],
            "result_count": c["result_count"], "second_round": c["range_search_second_round"]})
    json.dump(out, open(f"{HERE}/filtered_search.json", "w"), separators=(",", ":"))
    print("filtered cases:", {k: len(v) for k, v in out.items()})


def paged():
    out = []
    for path in sorted(glob.glob(f"{REF}/diskann/test/generated/graph/test/cases/paged_search/*.json")):
        c = json.load(open(path))["payload"]
        out.append({"source": os.path.relpath(path, REF), "name": os.path.basename(path)[:-5], "grid_dims": c["dims"],
                    "grid_size": c["grid_size"], "query": c["query"], "search_l": c["search_l"],
                    "page_size": c["page_size"], "pages": c["pages"], "total_results": c["total_results"],
                    # single_page asks for one page only (paged_search.rs:196-199)
                    "max_pages": 1 if os.path.basename(path) == "single_page.json" else None})
    json.dump(out, open(f"{HERE}/paged_search.json", "w"), separators=(",", ":"))
    print("paged cases:", len(out))


def f16_table():
    bits = np.zeros(65536, np.uint32)
    seen = np.zeros(65536, bool)
    for line in open(f"{REF}/diskann-wide/test_data/float16_conversion.txt"):
        h, v = line.strip().split(", ")
        h = int(h, 16)
        if v == "neg_infinity":
            f = float("-inf")
        elif v == "infinity":
            f = float("inf")
        elif v == "nan":
            f = float("nan")
        else:
            f = float(v)
        b = struct.unpack("<I", struct.pack("<f", np.float32(f)))[0]
        if h & 0x8000 and f == 0.0:  # "-0.0" parses with sign; keep the sign bit
            b |= 0x80000000
        bits[h] = b
        seen[h] = True
    assert seen.all()
    np.savez_compressed(f"{HERE}/f16_to_f32.npz", f32_bits=bits)
    print("f16 table written")


if __name__ == "__main__":
    grid_search()
    grid_insert()
    range_search()
    filtered()
    paged()
    f16_table()
