"""One-screen summary of a bench.py JSON line (file argument)."""
import json, sys
d = json.load(open(sys.argv[1]))
c = d["config"]
print(f"CFM     {d['value']:9.1f} tiles/s {d['ms_per_step']:7.3f} ms  conv {d['roofline']['achieved']:7.1f} TF ({d['roofline']['frac']:.4f})"
      f"  graph {c.get('graph_ms_per_step')} eager {c.get('eager_ms_per_step')}")
for k, v in d.get("kernels", {}).items():
    print("   ", k, v)
p = d.get("pix2pix")
if p:
    pc = p["config"]
    print(f"pix2pix {p['value']:9.1f} tiles/s {p['ms_per_step']:7.3f} ms  conv {p['roofline']['achieved']:7.1f} TF ({p['roofline']['frac']:.4f})"
          f"  graph {pc.get('graph_ms_per_step')} eager {pc.get('eager_ms_per_step')}  traffic {p['roofline'].get('traffic')}")
    for k, v in p.get("kernels", {}).items():
        print("   ", k, v)
if "sample" in d:
    s = d["sample"]
    print("sample  ", {b: {h: s[b][h]["ms_per_euler_step"] for h in s[b]} for b in ("batch32", "batch1")})
if "fp32_parity" in d:
    print("fp32    ", d["fp32_parity"]["value"], "tiles/s", d["fp32_parity"]["ms_per_step"], "ms")
if "cpu_baseline" in d:
    print("cpu     ", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "threads")
