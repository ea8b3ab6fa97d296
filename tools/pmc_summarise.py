"""Turn the two rocprofv3 --pmc passes of tools/pmc_traffic.sh into bytes per launch per kernel.

Counter units and gfx950 corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and
WRITE_SIZE are in KiB; WRITE_SIZE is exact for streaming stores; FETCH_SIZE reports half of
the bytes of a 16-byte-per-lane coalesced streaming read and is doubled for those kernels.
Calibrated here on kernels whose reads are known exactly (N = 60 000, 784 families):
  em_p1_fast<10,2>     16 B/lane loads, 12 float rows: expected 2258 MB, raw counter 1132 MB -> x2
  evidence_scatter      4 B/lane loads:                 expected  376 MB, raw counter  376 MB -> x1
  leaf_family_p1<..,2>  8 B/lane loads:                 expected  800 MB, raw counter  786 MB -> x1
so the doubling is applied only to kernels that load 16 bytes per lane.
Later calibration of the 8-byte-per-lane pattern of the current leaf-family kernel, on its
fresh-state instantiation whose reads are known exactly (2 evidence floats per family and 10
doubles of the parent product per observation: 382 MB at 60 160 padded observations):
  leaf_family_p1<10,2,2,3>  raw counter 214.5 MB -> x1.78
and with that factor the steady and reduce instantiations (which also read the 4-byte loop state)
land within 5 % of their known 571 MB, so x1.78 is used for all leaf_family_p1 instantiations."""
import csv, glob, json, re, sys
from collections import defaultdict

out_dir, dest = sys.argv[1], sys.argv[2]


def per_kernel(counter):
    path = glob.glob(f"{out_dir}/{counter}/*/*counter_collection.csv")[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")

# Calibration of the leaf-family kernel's 8-byte-per-lane pattern on its fresh-state instantiation
# (MODE 3 = loop mode, fresh state; template arguments <T, D0, S, V, MODE>) at the default bench
# geometry: 784 families x 60 160 padded observations x 2 evidence floats + 10 doubles of the
# parent product per observation.  The default EM step no longer launches that instantiation (its
# first iteration ingests the caller's evidence, MODE 9, with 16-byte loads): the factor measured
# when it did, 1.89, is kept unless the run at hand contains the instantiation again.
LEAF_FACTOR = 1.89
LEAF = re.compile(r"leaf_family_p1<(?:float|double), (\d+), (\d+), (\d+), (\d+)>")
for name, values in fetch.items():
    m = LEAF.search(name)
    if m and m.groups() == ("10", "2", "2", "3") and "float" in m.group(0) and values:
        known = 784 * 60160 * 8 + 10 * 60160 * 8
        LEAF_FACTOR = known / (1024.0 * sum(values) / len(values))
        print(f"leaf-family fetch calibration: x{LEAF_FACTOR:.3f}")
# The loop launches (leaf_loop_p1<T, D0, S, OUT, STATE>) bring the evidence in with 16-byte-per-lane
# LDS-DMA loads and read the parent product and the chunk state with 16-byte loads: the doubling
# is expected to apply.  Checked on the instantiation whose reads are known exactly at the default
# bench geometry (OUT 0, STATE 0: evidence + 10 doubles of the parent product per observation + 8
# chunks x (8 + 4) bytes of chunk state per observation).
LOOP = re.compile(r"leaf_loop_p1<(?:float|double), (\d+), (\d+), (\d+), (\d+)(?:, (?:true|false))?>")
LOOP_FACTOR = 2.0
for name, values in fetch.items():
    m = LOOP.search(name)
    if m and m.groups() == ("10", "2", "0", "0") and "float" in m.group(0) and values:
        known = 784 * 60160 * 8 + 10 * 60160 * 8 + 8 * 60160 * 12
        LOOP_FACTOR = known / (1024.0 * sum(values) / len(values))
        print(f"leaf-loop fetch calibration: x{LOOP_FACTOR:.3f}")
result = {}
for name in sorted(set(fetch) | set(write)):
    if "bn::" not in name:
        continue
    f = fetch.get(name, [0.0])
    w = write.get(name, [0.0])
    short = re.sub(r"\(.*", "", name).replace("void ", "")
    # kernels whose widest load is below 16 bytes per lane (leaf_family_p1<D0, S, V, MODE>: V floats per lane)
    narrow = (("evidence_scatter" in short and "pairs" not in short) or "ll_partial" in short or "generic" in short
              or "gather_rows" in short or "family_marginal" in short or "wide_combine" in short
              )
    factor = 1.0 if narrow else 2.0
    m = LEAF.search(short)
    if m:
        # the ingesting instantiation loads 16 bytes per lane from the caller's arrays (x2 like
        # every such kernel); the others read evidence and loop state 8 bytes per lane
        # (and so does the reduce-mode launch with four observations per lane)
        factor = 2.0 if int(m.group(4)) // 3 == 3 or m.group(3) == "4" else LEAF_FACTOR
    if LOOP.search(short):
        factor = LOOP_FACTOR
    result[short] = dict(
        launches=len(f),
        fetch_correction=factor,
        fetch_bytes_per_launch=factor * 1024.0 * sum(f) / len(f),
        write_bytes_per_launch=1024.0 * sum(w) / len(w),
        fetch_raw_kib=sum(f) / len(f),
    )
    result[short]["hbm_bytes_per_launch"] = result[short]["fetch_bytes_per_launch"] + result[short]["write_bytes_per_launch"]
json.dump(result, open(dest, "w"), indent=1)
for k, v in result.items():
    print(f"{k:60s} launches {v['launches']:4d}  fetch {v['fetch_bytes_per_launch'] / 1e6:10.1f} MB  write {v['write_bytes_per_launch'] / 1e6:10.1f} MB")
