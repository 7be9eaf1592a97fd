import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print(d["metric"], "|", d["value"], d["unit"], "|", d["ms_per_step"], "ms/step | conv", d["roofline"]["achieved"], "TF/s share", d["roofline"]["share_of_step_time"], "| fwd ms/tile", d["fwd_ms_per_tile"])
