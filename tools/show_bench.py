"""Pretty-print a bench.py JSON line: python tools/show_bench.py FILE [key ...]"""
import json
import sys


def show(d, ind=0, only=None):
    for k, v in d.items():
        if only and ind == 0 and k not in only:
            continue
        if isinstance(v, dict):
            print(" " * ind + k + ":")
            show(v, ind + 2)
        else:
            print(" " * ind + f"{k}: {str(v)[:140]}")


d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
show(d, only=set(sys.argv[2:]) or None)
