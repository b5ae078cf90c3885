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
    f16_table()
