import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"],d["ms_per_step"],d["ms_per_step_mean"],d["ms_per_step_min"],d["ms_per_step_max"])
print({k:r.get(k) for k in ("bound","kernel","achieved","peak","unit","frac","traffic","clock_mhz","step_frac","kernel_times_sum_ms","profiled_step_ms","event_pair_overhead_ms")})
for k,v in r["all_mfma_kernels"].items(): print(k,{a:b for a,b in v.items() if a not in ("peak","stash_bytes_per_launch")})
print(r["other_kernels_ms"])
if d.get("config3"): print("config3", d["config3"]["value"], d["config3"]["ms_per_step"], {k:v["avg_ms"] for k,v in d["config3"]["roofline"]["all_mfma_kernels"].items()})
if d.get("cpu_baseline"): print("cpu", d["cpu_baseline"]["value"])
