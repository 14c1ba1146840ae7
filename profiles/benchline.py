"""Prints a one-line digest of bench.py JSON lines.  Usage: benchline.py FILE... (or JSON lines on stdin)."""
import json
import sys


def digest(lines):
    for line in lines:
        line = line.strip()
        if line.startswith("{"):
            d = json.loads(line)
            c, r = d["config"], d["roofline"]
            print("%.2f Mframes/s  kernel_ms=%.2f  mean_iters=%.2f conv=%.3f  achieved=%.1f GB/s  %s"
                  % (d["value"] / 1e6, r["kernel_ms"], c["mean_iterations_run"], c["converged_fraction"], r["achieved"],
                     r["kernel"]))
            for k, v in (d.get("secondary") or {}).items():
                print("   %-55s %8.1f Mframes/s" % (k, v["frames_per_s"] / 1e6))


if len(sys.argv) > 1:
    for path in sys.argv[1:]:
        with open(path) as f:
            digest(f)
else:
    digest(sys.stdin)
