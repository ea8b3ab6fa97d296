"""Stamp a counter summary (pmc_traffic / pmc_valu / pmc_mfma JSON) with where it comes from:

    python3 tools/pmc_meta.py <bench line JSON> <summary JSON> [<summary JSON> ...]

adds "_meta": {"git": $BN_GIT_HEAD, "signatures": {profile label: register / LDS / spill signature of
its kernels}} taken from the `kernels` (and `secondary[].roofline`) of a bench.py line of the same
build.  bench.py hands a summary's figures out only while the running kernels still have the
recorded signature."""
import json
import os
import sys

line = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]
bench = json.loads(line)
signatures = {k["kernel"]: k.get("signature") for k in bench.get("kernels", [])}
for entry in bench.get("secondary", []) or []:
    roof = entry.get("roofline") or {}
    if roof.get("kernel"):
        signatures.setdefault(roof["kernel"], roof.get("signature"))  # (a label of the line's own kernels stays theirs)
meta = dict(git=os.environ.get("BN_GIT_HEAD"), signatures=signatures)
for path in sys.argv[2:]:
    data = json.load(open(path))
    data["_meta"] = meta
    json.dump(data, open(path, "w"), indent=1)
