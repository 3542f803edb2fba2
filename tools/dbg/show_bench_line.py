import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v.get("value") for k, v in d.get("variants", {}).items()}, d["cpu_baseline"]["value"], d["roofline"].get("valu_pipe_frac"), d["roofline"].get("frac"))
