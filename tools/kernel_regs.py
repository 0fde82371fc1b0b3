#!/usr/bin/env python3
"""Per-kernel resource usage (VGPRs, SGPRs, spills, scratch, LDS) of the gfx950 code objects inside a built library.
usage: tools/kernel_regs.py [path/to/librl_render.so] [name filter regex]        (CPU only: reads the ELF notes with llvm-readelf)"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "rendering-learning_amd", "csrc", "librl_render.so")
flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"
data = open(lib, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
i = 0
rows = []
while True:
    i = data.find(magic, i)
    if i < 0:
        break
    ne = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(ne):
        o, s, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tl].decode()
        off += tl
        if "gfx950" not in triple or s == 0:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(data[i + o:i + o + s])
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], stdout=subprocess.PIPE, text=True).stdout
        cur = {}
        for line in txt.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == "name" and v.startswith("_Z") or (k == "name" and "kernel" in v and not cur.get("symbol")):
                cur["name"] = v
            if k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count", "symbol"):
                cur[k] = v
            if k == "wavefront_size":
                if "name" in cur:
                    rows.append(cur)
                cur = {}
    i += 24
names = [r["name"] for r in rows]
dem = subprocess.run([CXXFILT], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.splitlines() if names else []
print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'v_spill':>7} {'s_spill':>7} {'scratch':>8} {'lds':>7}  kernel")
for r, d in zip(rows, dem):
    d = re.sub(r"^void ", "", d).replace("rl::", "")
    d = re.sub(r"\(.*\)$", "", d)[:110]
    if flt and not flt.search(d):
        continue
    print(f"{r.get('vgpr_count', '?'):>5} {r.get('agpr_count', '0'):>5} {r.get('sgpr_count', '?'):>5} {r.get('vgpr_spill_count', '0'):>7} {r.get('sgpr_spill_count', '0'):>7} "
          f"{r.get('private_segment_fixed_size', '0'):>8} {r.get('group_segment_fixed_size', '0'):>7}  {d}")
