"""Summary of a bench.py JSON line: python tools/show_bench.py [file]  (stdin when no file)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read())
r = d.get("roofline") or {}
c = r.get("cells", {})
print("%s | B=%s %.0f seg/s %.3f ms elbo %.3f | frac %.4f | %s" % (
    d["config"]["workload"][:2], d["config"]["global_batch"], d["value"], d["ms_per_step"], d["elbo_nats_per_frame"], r.get("frac", 0.0),
    {k: (round(v["avg_launch_us"], 1), round(v["tflops"], 1)) for k, v in c.items()}))
for k, v in (r.get("op_ms_per_step") or {}).items():
    print("   %-36s %.4f ms" % (k, v))
for a in d.get("alt", []):
    print("   alt: %-80s %.0f seg/s %.3f ms" % (a["workload"][-60:], a["value"], a["ms_per_step"]))
if "idx" in d:
    print("   idx:", d["idx"])
if "cpu_baseline" in d:
    cb = d["cpu_baseline"]
    print("   cpu: %.1f seg/s, %d threads, median %.0f ms, elbo cpu %.5f gpu-f32 %s" % (
        cb["value"], cb.get("threads", 0), cb.get("median_ms", 0.0), cb.get("elbo_nats_per_frame", 0.0), cb.get("gpu_f32_elbo_nats_per_frame")))
