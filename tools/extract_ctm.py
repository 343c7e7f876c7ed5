#!/usr/bin/env python3
"""Pack the continuum absorption coefficient tables into one binary blob.

The CO2/H2O continua (3 x 2001 values each) and the N2/O2 collision-induced
absorption tables (2 x 98, 2 x 90 values) are physical data the forward model
needs (reference src/ctmco2.tbl, ctmh2o.tbl, ctmn2.tbl, ctmo2.tbl, consumed at
jr_common.h:315-390).  This script reads the numbers from the reference tree
(only available in the build container) and writes them as little-endian
float64 into jurassic-gpu_amd/data/ctm.bin, in this fixed order:

    co2296[2001] co2260[2001] co2230[2001]
    h2o296[2001] h2o260[2001] h2ofrn[2001]
    n2_b[98] n2_beta[98] o2_b[90] o2_beta[90]

The blob (67 kB) is committed; library and oracle .incbin it.
"""
import re, struct, sys, hashlib
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src")
OUT = Path(sys.argv[2]) if len(sys.argv) > 2 else Path(__file__).resolve().parent.parent / "jurassic-gpu_amd" / "data" / "ctm.bin"

def arrays(fname):
    txt = (REF / fname).read_text()
    out = {}
    for m in re.finditer(r"\((\w+)\)\s*\[(\d+)\]\s*=\s*\{([^}]*)\}", txt):
        vals = [float(v) for v in m.group(3).replace("\n", " ").split(",") if v.strip()]
        assert len(vals) == int(m.group(2)), (fname, m.group(1), len(vals))
        out[m.group(1)] = vals
    return out

co2, h2o, n2, o2 = (arrays(f) for f in ("ctmco2.tbl", "ctmh2o.tbl", "ctmn2.tbl", "ctmo2.tbl"))
seq = [co2["co2296"], co2["co2260"], co2["co2230"],
       h2o["h2o296"], h2o["h2o260"], h2o["h2ofrn"],
       n2["ba"], n2["betaa"], o2["ba"], o2["betaa"]]
assert [len(s) for s in seq] == [2001] * 6 + [98, 98, 90, 90]
blob = b"".join(struct.pack("<%dd" % len(s), *s) for s in seq)
OUT.write_bytes(blob)
print(OUT, len(blob), "bytes sha256", hashlib.sha256(blob).hexdigest())
