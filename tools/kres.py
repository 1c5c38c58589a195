"""register / scratch / occupancy table of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage):
python tools/kres.py point-cloud-reid_amd/csrc/train_chain_kernels.hip [name filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = ["-ffp-contract=off"] if os.path.basename(src) in ("point_ops.hip", "edge_kernels.hip") else []
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(ROOT, "point-cloud-reid_amd", "csrc"), "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra + os.environ.get("PCR_EXTRA_HIPCC_FLAGS", "").split()
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"),
                     ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("sgpr", r" SGPRs: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print("%5s %5s %6s %8s %4s  %s" % ("vgpr", "agpr", "spill", "scratch", "occ", "kernel"))
for r in rows:
    if flt in r["name"]:
        nm = re.sub(r"\(anonymous namespace\)::", "", r["name"])
        print("%5s %5s %6s %8s %4s  %s" % (r.get("vgpr"), r.get("agpr"), r.get("spill"), r.get("scratch"), r.get("occ"), nm[:150]))
