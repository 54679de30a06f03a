import sys, json
d = json.loads(sys.stdin.read())
c = d["roofline"]["cells"] if "roofline" in d and d["roofline"] else {}
print("B=%s %.0f seg/s %.3f ms elbo %.3f | %s" % (d["config"]["global_batch"], d["value"], d["ms_per_step"], d["elbo_nats_per_frame"],
      {k: (round(v["avg_launch_us"], 1), round(v["tflops"], 1)) for k, v in c.items()}))
