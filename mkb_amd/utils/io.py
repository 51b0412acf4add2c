"""CSV / JSON readers with the reference's return types (mkb/utils/read_csv.py:8-22, read_json.py:6-8)."""
import csv
import json

__all__ = ["read_csv", "read_json"]


def read_csv(file_path):
    with open(f"{file_path}", "r") as f:
        return [(int(h), int(r), int(t)) for h, r, t in csv.reader(f)]


def read_json(file_path):
    with open(file_path, "r") as f:
        return json.load(f)
