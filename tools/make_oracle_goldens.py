#!/usr/bin/env python3
"""Writes tests/golden/oracle_examples.json: the oracle's radiances, transmittances, tangent points and LOS
point counts for the two reference examples (shipped atm.tab / obs geometry of rad.org) with the seeded
synthetic emissivity tables of jurassic_hip/synth.py.

These are NOT reference-produced numbers (the reference's tables are missing blobs): they
pin the oracle against unnoticed changes and give the GPU tests a committed target.  Doubles are stored as
C99 hex strings (exact)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import common
from oracle import orc

orc.build()
out = {"generator": "tools/make_oracle_goldens.py", "tables": "jurassic_hip.synth.table_rows defaults (33 p x 10 T, ratio 1.122)",
       "cases": {}}
for name, case in (("limb", common.limb_case()), ("nadir", common.nadir_case()),
                   ("limb_four_continua", common.limb_case(nu=common.CTM4_NU))):
    ref = orc.formod_rays(case.ctl, case.atm, case.oracle_tables(orc), case.geom)
    out["cases"][name] = {
        "emitters": [e.decode() if isinstance(e, bytes) else str(e) for e in common.LIMB_EMITTERS] if name != "nadir" else ["CO2"],
        "rays": int(len(case.geom)),
        "np": [int(x) for x in ref["np"]],
        "rad": [[float(v).hex() for v in row] for row in ref["rad"]],
        "tau": [[float(v).hex() for v in row] for row in ref["tau"]],
        "tp": [[float(v).hex() for v in row] for row in ref["tp"]],
    }
path = os.path.join(ROOT, "tests", "golden", "oracle_examples.json")
open(path, "w").write(json.dumps(out, separators=(",", ":")).replace('"cases":{', '"cases":{\n').replace('},"', '},\n"') + "\n")
print(path, os.path.getsize(path), "bytes")
