#!/bin/bash
# Samples socket power and shader clock (rocm-smi) every 0.2 s while a command runs; prints mean / max of the samples taken
# after the first 3 s.   scripts/power_watch.sh <label> <command ...>
label=$1; shift
out=$(mktemp)
( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n' >> "$out"; echo >> "$out"; sleep 0.2; done ) &
wpid=$!
e0=$(rocm-smi --showenergycounter --json 2>/dev/null | tr -d '\n')
t0=$(date +%s.%N)
"$@"
rc=$?
t1=$(date +%s.%N)
e1=$(rocm-smi --showenergycounter --json 2>/dev/null | tr -d '\n')
kill $wpid 2>/dev/null
python3 - "$e0" "$e1" "$t0" "$t1" <<'PY'
import json, sys
try:
    a, b = (json.loads(x)["card0"] for x in sys.argv[1:3])
    k = [k for k in a if "ccumulated" in k or "nergy" in k]
    ja = float(a[[x for x in k if "uJ" in x or "Accumulated" in x][0]]); jb = float(b[[x for x in k if "uJ" in x or "Accumulated" in x][0]])
    dt = float(sys.argv[4]) - float(sys.argv[3])
    print(f"[power_watch] energy {((jb - ja) / 1e6):.1f} J over {dt:.1f} s wall = {(jb - ja) / 1e6 / dt:.0f} W mean (whole process, incl. start-up); keys {k}")
except Exception as e:
    print("[power_watch] no energy counter:", e, sys.argv[1][:200])
PY
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
