import re
import subprocess
import sys

cur, rows = None, {}
for l in sys.stdin:
    m = re.search(r"remark: +(.*?) \[-Rpass", l)
    if not m:
        if "error" in l:
            print(l.rstrip())
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0][:72]
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    print(f"{name:74s} vgpr {v.get('VGPRs', '?'):>4} agpr {v.get('AGPRs', '?'):>3} sgpr {v.get('SGPRs', '?'):>3} "
          f"spill {v.get('VGPRs Spill', '?')} scratch {v.get('ScratchSize [bytes/lane]', '?')} occ {v.get('Occupancy [waves/SIMD]', '?')}")
