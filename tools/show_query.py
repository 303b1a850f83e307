#!/usr/bin/env python
# coding: utf-8
"""One line per metric of bench_query.py's JSON lines (stdin or files)."""
import json
import sys

for src in ([open(a) for a in sys.argv[1:]] or [sys.stdin]):
    for line in src:
        try:
            d = json.loads(line)
        except Exception:
            continue
        print(f"{d['metric'][:64]:64s} {d['value']:.4g} {d.get('unit', '')}")
