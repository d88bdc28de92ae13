"""Registers / scratch / LDS per kernel of libsmr_hip.so from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python scripts/kernel_resources.py [remarks.txt]  (without a file: compiles csrc/smr_hip.hip, ~1 min)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    text = open(sys.argv[1]).read()
else:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-o", "/dev/null"] + os.environ.get("SMR_HIPCC_FLAGS", "").split() + [
                        os.path.join(ROOT, "safemotionsrisk_amd", "csrc", "smr_hip.hip"), "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    text = p.stderr
cur, rows = None, {}
for line in text.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?): (\d+)", line) or re.search(r":\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-62s %6s %6s %8s %8s %6s" % ("kernel", "VGPR", "AGPR", "spill", "scratch", "LDS"))
for k, r in rows.items():
    g = lambda *names: next((r[n] for n in r for nm in names if n.startswith(nm)), 0)
    print("%-62s %6d %6d %8d %8d %6d" % (k[:62], g("VGPRs:") or g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g("ScratchSize"), g("LDS Size")))
