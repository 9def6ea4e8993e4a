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
for n, off in enumerate(outs):
    end = outs[n+1] if n + 1 < len(outs) else len(data)
    f = tempfile.NamedTemporaryFile(suffix=".co", delete=False); f.write(data[off:end]); f.close()
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True)
    os.unlink(f.name)
    cur = {}
    for line in r.stdout.splitlines():
        m = re.match(r"\s*-?\s*\.(name|vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s*(\S+)", line)
        if m:
            if m.group(1) == "name" and "name" in cur and "vgpr_count" in cur:
                pass
            cur[m.group(1)] = m.group(2)
        if line.strip().startswith("- .agpr_count") or line.strip().startswith("- .args"):
            if "name" in cur and "vgpr_count" in cur and pat in cur["name"]:
                print(cur)
            cur = {} if line.strip().startswith("- .a") else cur
