#!/usr/bin/env python3
"""Pack the mid-latitude climatology profiles into one binary blob.

The reference's `climatology` tool (src/climatology.c, src/jurassic.c:79-140)
interpolates 121-level profiles of pressure, temperature and 27 trace gases
(src/climatology.tbl: physical data, one C initialiser per quantity).  This
script reads the numbers from the reference tree (only available in the build
container) and writes jurassic-gpu_amd/data/clim.bin:

    repeated { char name[8] (lower case, NUL padded); float64 values[121] }

in file order (z, pre, tem, then the gases).  The blob (30 kB) is committed;
the `climatology` executable .incbin's it.
"""
import re, struct, sys, hashlib
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src")
OUT = Path(sys.argv[2]) if len(sys.argv) > 2 else Path(__file__).resolve().parent.parent / "jurassic-gpu_amd" / "data" / "clim.bin"

txt = (REF / "climatology.tbl").read_text()
blob = b""
names = []
for m in re.finditer(r"(\w+)\s*\[(\d+)\]\s*=\s*\{([^}]*)\}", txt):
    name, n = m.group(1).lower(), int(m.group(2))
    vals = [float(v) for v in m.group(3).replace("\n", " ").split(",") if v.strip()]
    assert n == 121 and len(vals) == 121, (name, n, len(vals))
    assert len(name) <= 7
    blob += name.encode().ljust(8, b"\0") + struct.pack("<121d", *vals)
    names.append(name)
OUT.write_bytes(blob)
print(OUT, len(names), "profiles:", " ".join(names))
print(len(blob), "bytes sha256", hashlib.sha256(blob).hexdigest())
