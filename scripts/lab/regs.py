"""Register / LDS / occupancy table of every kernel in a HIP source (hipcc -Rpass-analysis=kernel-resource-usage), CPU-only.
usage: python scripts/lab/regs.py file.hip [substring filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", "nabladft_amd/csrc", src, "--cuda-device-only", "-c", "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
    if "error" in line:
        print(line)
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:110]:110s} V{r.get('VGPRs',0):4d} A{r.get('AGPRs',0):4d} occ{r.get('Occupancy [waves/SIMD]',0):2d} spill{r.get('VGPRs Spill',0):4d} scratch{r.get('ScratchSize [bytes/lane]',0):5d} lds{r.get('LDS Size [bytes/block]',0):7d}")
