"""Per-kernel register / LDS report of one HIP source: python tools/kres.py file.hip [name-filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-c", "--cuda-device-only", src,
                      "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print(k[:90], " ".join(f"{a}={v.get(b)}" for a, b in (("V", "VGPRs"), ("A", "AGPRs"), ("Vsp", "VGPRs Spill"), ("Ssp", "SGPRs Spill"),
                                                               ("scr", "ScratchSize"), ("occ", "Occupancy"), ("lds", "LDS Size"))))
