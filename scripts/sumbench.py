import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"],2), round(d["roofline"]["achieved"],1), {k: round(v,2) for k,v in d["kernel_ms_per_step"].items()})
