import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "walk", d["value_with_every_primary_ray_walking_the_tree"], d["kernel_ms_per_step"])
print({k: r.get(k) for k in ("frac", "useful_lane_frac", "issue_frac", "arithmetic_frac", "lane_utilisation", "achieved", "peak", "traffic", "issue_frac_at_that_clock", "clock_GHz_during_k_trace", "boxes_per_ray", "boxes_per_shadow_ray", "avg_launch_ms", "launches")})
print("hbm", r["hbm"]["frac"], r["hbm"].get("traffic_GBps"), "shade", {k: v for k, v in r["shade"].items() if k != "note"})
print("vm", r.get("vector_memory"))
print("cpu", d["cpu_baseline"])
print("config", d["config"]["accumulations_per_batch"], d["config"]["accumulations_per_batch_limit"])
