"""Pack the benchmark DATA files (h,r,t integer CSVs + label JSONs) that the BASELINE configs
need into compact ``.npz`` assets under ``mkb_amd/datasets/data/``.

Run in the build container only (reads ``/root/reference/mkb/datasets/<name>/``, which does not
exist on the GPU box).  Data, not code: public benchmark triples (CountriesS1, Umls, WN18RR,
FB15k-237, YAGO3-10 valid/test).  ``yago310/train.csv`` is absent from the reference mount
(``.MISSING_LARGE_BLOBS``); ``mkb_amd.datasets.Yago310`` synthesises its training triples.
"""
import csv
import json
import pathlib
import sys

import numpy as np

SRC = pathlib.Path("/root/reference/mkb/datasets")
DST = pathlib.Path(__file__).resolve().parent.parent / "mkb_amd" / "datasets" / "data"
NAMES = ["countries_s1", "umls", "wn18rr", "fb15k237", "yago310"]


def read_triples(path):
    if not path.exists():
        return np.zeros((0, 3), dtype=np.int32)
    with open(path) as f:
        rows = [(int(h), int(r), int(t)) for h, r, t in csv.reader(f)]
    return np.asarray(rows, dtype=np.int32).reshape(-1, 3)


def labels_in_id_order(path):
    with open(path) as f:
        d = json.load(f)
    inv = sorted(d.items(), key=lambda kv: kv[1])
    assert [i for _, i in inv] == list(range(len(inv))), "ids must be dense 0..n-1"
    return json.dumps([k for k, _ in inv], ensure_ascii=False)


def main():
    DST.mkdir(parents=True, exist_ok=True)
    for name in NAMES:
        p = SRC / name
        out = {
            "train": read_triples(p / "train.csv"),
            "valid": read_triples(p / "valid.csv"),
            "test": read_triples(p / "test.csv"),
            "entities": np.frombuffer(labels_in_id_order(p / "entities.json").encode("utf-8"), dtype=np.uint8),
            "relations": np.frombuffer(labels_in_id_order(p / "relations.json").encode("utf-8"), dtype=np.uint8),
        }
        np.savez_compressed(DST / f"{name}.npz", **out)
        print(name, {k: v.shape for k, v in out.items()}, (DST / f"{name}.npz").stat().st_size, file=sys.stderr)


if __name__ == "__main__":
    main()
