"""Prints the headline and the per-kernel extras of a bench.py JSON line (file argument)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", round(d["value"], 2), d["unit"], "frac", round(d["roofline"]["frac"], 3))
for k, v in d["extra"].items():
    print(" ", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "kernels"})
