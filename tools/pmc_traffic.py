"""Derive per-launch HBM traffic of the two roofline kernels from a rocprofv3 PMC summary (tools/summarize_prof.py output).
FETCH_SIZE is doubled (gfx950 counts 128-B reads as 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is.
usage: python tools/pmc_traffic.py profiles/r01_rocprof_summary.txt > profiles/r01_pmc_traffic.json"""
import json
import re
import sys

txt = open(sys.argv[1]).read()
sec = {}
cur = None
for line in txt.splitlines():
    if line.startswith("== rocprofv3 --pmc FETCH_SIZE") or line.startswith("== rocprofv3 --pmc WRITE_SIZE"):
        cur = "FETCH" if "--pmc FETCH_SIZE" in line else "WRITE"
        sec[cur] = {}
    elif line.startswith("== "):
        cur = None
    elif cur and line and not line.startswith("kernel"):
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            sec[cur][m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))   # dispatches, mean KiB


def kib(name, which):
    return sec[which].get(name, (0, 0.0))


# Desc stack = the 8 Cylindrical_Net layers (layers 1 and 5 share one instantiation: its mean counts twice)
desc = [k for k in sec["FETCH"] if (k.startswith("conv_kernel<") and ", 140, 198, 140," in k) or k.startswith("wino_kernel<") or k.startswith("wino_pair_kernel<") or k.startswith("wino43_kernel<")]
tot = 0.0
detail = {}
for k in desc:
    mult = 2 if (k.startswith("conv_kernel<4, 9, 140, 198, 140, 64,") or k.startswith("wino_kernel<4, 64,") or k.startswith("wino_pair_kernel<4, 64,") or k.startswith("wino43_kernel<4, 64,")) else 1   # layers 1 and 5 share an instantiation
    b = (2.0 * kib(k, "FETCH")[1] + kib(k, "WRITE")[1]) * 1024.0
    detail[k] = {"fetch_KiB_raw": kib(k, "FETCH")[1], "write_KiB": kib(k, "WRITE")[1], "bytes": b, "layers": mult}
    tot += mult * b
bq = [k for k in sec["FETCH"] if k.startswith("ball_query_kernel")]
n = sum(kib(k, "FETCH")[0] for k in bq)
bqb = sum(kib(k, "FETCH")[0] * (2.0 * kib(k, "FETCH")[1] + kib(k, "WRITE")[1]) * 1024.0 for k in bq) / max(n, 1)
print(json.dumps({"source": sys.argv[1], "fetch_correction": "FETCH_SIZE x2 (gfx950)",
                  "desc_conv_stack_bytes_per_launch": tot, "desc_conv_layers": detail,
                  "ball_query_bytes_per_launch": bqb, "ball_query_dispatches": n}, indent=1))
