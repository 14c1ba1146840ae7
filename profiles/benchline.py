import json,sys
for line in sys.stdin:
    line=line.strip()
    if line.startswith("{"):
        d=json.loads(line); print("%.2f Mframes/s  kernel_ms=%.2f  %s" % (d["value"]/1e6, d["roofline"]["kernel_ms"], d["roofline"]["kernel"]))
