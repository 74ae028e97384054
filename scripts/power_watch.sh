#!/bin/bash
# Samples socket power and shader clock (rocm-smi) every 0.2 s while a command runs; prints mean / max of the samples taken
# after the first 3 s.   scripts/power_watch.sh <label> <command ...>
label=$1; shift
out=$(mktemp)
( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n' >> "$out"; echo >> "$out"; sleep 0.2; done ) &
wpid=$!
"$@"
rc=$?
kill $wpid 2>/dev/null
python3 - "$label" "$out" <<'PY'
import json, re, sys
label, path = sys.argv[1], sys.argv[2]
pw, ck = [], []
for i, line in enumerate(open(path)):
    if i < 15 or not line.strip():
        continue
    try:
        d = json.loads(line)
    except Exception:
        continue
    c = d.get("card0", {})
    for k, v in c.items():
        if "ower" in k and "W" in k:
            try: pw.append(float(v))
            except Exception: pass
        if k.startswith("sclk clock speed"):
            m = re.search(r"(\d+)", str(v))
            if m: ck.append(float(m.group(1)))
def st(x): return f"mean {sum(x)/len(x):.0f} max {max(x):.0f} n={len(x)}" if x else "n/a"
print(f"[power_watch] {label}: power W {st(pw)} | sclk MHz {st(ck)}")
PY
rm -f "$out"
exit $rc
