"""Generate tests/golden/lj_eos_T3.json from the REFERENCE's own equation-of-state tool.

oracle/_ref/lj_eos is /root/reference/test/MD/tools/lj_eos.cpp compiled by oracle/ref.mk (g++ -O3, as
test/MD/tools/eos.sh:6).  This script drives it the way eos.sh:8-13 does (temperature, then density on
stdin; E = residual internal energy + 3/2 T, P = total pressure) over the density sweep of
test/MD/test.bash:28 (seq 0.1 0.05 1.0) at the test's temperature T = 3 (test.bash:4).
Run in the build container (needs /root/reference); the JSON travels, the binary's source does not.
"""
import json
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "ref.mk"], check=True)
EXE = os.path.join(ROOT, "oracle", "_ref", "lj_eos")

T = 3.0
rows = []
for i in range(19):
    rho = round(0.1 + 0.05 * i, 2)
    out = subprocess.run([EXE], input=f"{T}\n{rho}\n", capture_output=True, text=True).stdout
    P = float(re.search(r"Pressure:\s+(\S+)", out).group(1))
    U = float(re.search(r"Internal Energy:\s+(\S+)", out).group(1))
    rows.append({"rho": rho, "T": T, "E": U + 1.5 * T, "P": P})
doc = {
    "source": "reference test/MD/tools/lj_eos.cpp via test/MD/tools/eos.sh (LJ truncated+shifted, rc=2.5 sigma)",
    "config": {"numberParticles": 16384, "cutOff": 2.5, "sigma": 1, "epsilon": 1, "dt": 0.0005, "friction": 1.0,
               "integrator": "VerletNVT", "citation": "test/MD/test.bash:3-9,39-58"},
    "columns": "rho T E/N(total: kinetic + shifted potential) P(virial, truncated force)",
    "rows": rows,
}
with open(os.path.join(HERE, "lj_eos_T3.json"), "w") as f:
    json.dump(doc, f, indent=1)
print(json.dumps(rows[10], indent=1))
