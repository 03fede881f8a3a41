"""summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: python tools/kernel_regs.py remarks.txt [name-substring ...]"""
import re, sys
txt = open(sys.argv[1]).read()
subs = sys.argv[2:]
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()
    if subs and not all(s in name for s in subs):
        continue
    def f(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    print(name[:110], "| VGPR", f("VGPRs"), "AGPR", f("AGPRs"), "SGPR", f("SGPRs"), "scratch", f(r"ScratchSize \[bytes/lane\]"), "occ", f(r"Occupancy \[waves/SIMD\]"), "LDS", f(r"LDS Size \[bytes/block\]"))
