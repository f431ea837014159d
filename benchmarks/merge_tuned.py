#!/usr/bin/env python3
"""Merge first-use GEMM decisions dumped by ACT_GEMM_TUNE_SAVE=<dir>/tuned_%p.json (one file per process of a GPU test session) into the shipped
table act_amd/gemm_tune_gfx950.json: only keys the shipped table does not have are added (their value is the bit-stable first-use pick, so adding
them cannot change a result bit -- it only stops the suite from timing them again).

    python benchmarks/merge_tuned.py gpurun_out/tuned_*.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "act_amd", "gemm_tune_gfx950.json")


def main(files):
    t = json.load(open(TABLE))
    cfg = t["configs"]
    added, conflicts = {}, 0
    for f in files:
        for k, v in json.load(open(f))["configs"].items():
            if k in cfg:
                continue
            if k in added and added[k] != v:
                conflicts += 1                                    # two processes timed the same shape and preferred different (bit-identical) tiles
                continue
            added[k] = v
    cfg.update(added)
    t["configs"] = dict(sorted(cfg.items(), key=lambda kv: tuple(int(x) for x in kv[0].split(","))))
    if added and "test-suite shapes" not in t.get("note", ""):
        t["note"] = t.get("note", "") + "  Round 5: the shapes of the GPU test suite (tiny / odd geometries) added with their bit-stable first-use picks."
    json.dump(t, open(TABLE, "w"), indent=0)
    print(f"added {len(added)} shape(s) ({conflicts} differing picks among bit-identical candidates ignored); table now has {len(cfg)}")


if __name__ == "__main__":
    main(sys.argv[1:])
