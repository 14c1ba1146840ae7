import json, sys
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line); c = d["config"]; r = d["roofline"]
        print("%.2f Mframes/s  kernel_ms=%.2f  mean_iters=%.2f conv=%.3f  achieved=%.1f GB/s  %s"
              % (d["value"] / 1e6, r["kernel_ms"], c["mean_iterations_run"], c["converged_fraction"], r["achieved"], r["kernel"]))
