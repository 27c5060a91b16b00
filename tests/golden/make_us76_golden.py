"""Transcribes the 21-node U.S. Standard Atmosphere table the reference interpolates (src/sasktran2/climatology/us76.py:8-85: altitudes [m],
temperatures [deg C], pressures [1e4 Pa]) into tests/golden/us76_nodes.json -- numbers only, read from the reference's source as text."""
import ast
import json
import re
import sys
from pathlib import Path

src = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/sasktran2/climatology/us76.py").read_text()
out = {}
for name in ("_ALTS", "_TEMPERATURE_C", "_PRESSURE"):
    m = re.search(name + r"\s*=\s*np\.array\(\s*(\[.*?\])\s*\)", src, re.S)
    out[name.strip("_").lower()] = [float(x) for x in ast.literal_eval(m.group(1))]
assert len(out["alts"]) == len(out["temperature_c"]) == len(out["pressure"]) == 21
Path(__file__).with_name("us76_nodes.json").write_text(json.dumps(
    {"source": "src/sasktran2/climatology/us76.py:8-85", "altitude_m": out["alts"], "temperature_c": out["temperature_c"],
     "pressure_1e4_pa": out["pressure"]}, indent=1))
print("wrote", len(out["alts"]), "nodes")
