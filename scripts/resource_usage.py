#!/usr/bin/env python3
"""CPU: compile surfdisp_kernels.hip with -Rpass-analysis=kernel-resource-usage and print one line per kernel
instantiation (VGPRs, SGPRs, scratch, occupancy, LDS)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "pysurfinv_amd", "csrc")
flags = "-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize".split() + sys.argv[2:]
src = sys.argv[1] if len(sys.argv) > 1 else "surfdisp_kernels.hip"
r = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-I" + os.path.join(ROOT, "include"), "-I" + CS,
                    "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CS, src), "-o", "/dev/null"],
                   capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for d in rows:
    name = subprocess.run(["c++filt", d["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void sd::", "")
    print(f"{name:58s} VGPR {d.get('VGPRs','?'):>4s} AGPR {d.get('AGPRs','?'):>3s} SGPR {d.get('TotalSGPRs', d.get('SGPRs','?')):>4s} "
          f"scratch {d.get('ScratchSize [bytes/lane]','?'):>4s} occ {d.get('Occupancy [waves/SIMD]','?'):>2s} LDS {d.get('LDS Size [bytes/block]','?')}")
