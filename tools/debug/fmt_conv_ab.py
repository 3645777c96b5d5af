import sys,ast
for l in sys.stdin:
    try: d=ast.literal_eval(l)
    except Exception: print(l.strip()); continue
    if "name" in d and "fwd_us_f32" in d: print(d["name"], d["C"], d["K"], d["H"], d["k"], d["stride"], "fwd", d["fwd_us_f32"], d["fwd_us_b3"], "dgrad", d["dgrad_us_f32"], d["dgrad_us_b3"])
    else: print(d)
