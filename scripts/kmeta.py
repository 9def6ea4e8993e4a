#!/usr/bin/env python3
"""kernel resource usage from a shared object / object with embedded gfx950 code objects: python kmeta.py lib.so [pattern]"""
import re, subprocess, sys, os, tempfile
path = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
data = open(path, "rb").read()
# embedded code objects: ELF magic with AMDGPU machine (e_machine 0xE0)
outs = []
i = 0
while True:
    i = data.find(b"\x7fELF\x02\x01\x01", i)
    if i < 0: break
    if data[i+18:i+20] == b"\xe0\x00":
        outs.append(i)
    i += 4
KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size")
for n, off in enumerate(outs):
    end = outs[n+1] if n + 1 < len(outs) else len(data)
    f = tempfile.NamedTemporaryFile(suffix=".co", delete=False); f.write(data[off:end]); f.close()
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True)
    os.unlink(f.name)
    # one kernel = the lines between two "- .agpr_count" / "- .args" list heads inside amdhsa.kernels
    blocks, cur = [], None
    for line in r.stdout.splitlines():
        if re.match(r"\s*- \.(agpr_count|args):", line):
            if cur: blocks.append(cur)
            cur = {}
        if cur is None: continue
        m = re.match(r"\s*-?\s*\.(\w+):\s*(\S+)\s*$", line)
        if m and (m.group(1) in KEYS or (m.group(1) == "name" and m.group(2).startswith("_Z") or m.group(1) == "name" and "name" not in cur and not m.group(2).startswith("a"))):
            if m.group(1) == "name" and "kname" in cur: continue
            cur["kname" if m.group(1) == "name" else m.group(1)] = m.group(2)
        if line.startswith("amdhsa.target") and cur: blocks.append(cur); cur = None
    if cur: blocks.append(cur)
    for b in blocks:
        if "vgpr_count" in b and pat in b.get("kname", ""):
            print(b.get("kname"), " ".join(f"{k.replace('_count','').replace('_fixed_size','')}={b.get(k)}" for k in KEYS if k in b))
