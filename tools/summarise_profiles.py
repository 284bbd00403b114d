#!/usr/bin/env python3
"""Copies the summaries of one tools/profile_round.sh run from gpurun_out/ (scratch) into profiles/ (tracked) and
refreshes profiles/traffic.json.   usage: tools/summarise_profiles.py r02b"""
import collections, csv, json, os, shutil, statistics, sys
tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
for name in ("bench.json", "kernel_stats.csv", "pmc_fetch_write.csv", "pmc_units.txt", "init.txt", "table_stats.txt",
             "dither_knock.txt", "dither_stats.txt", "lab_rate.txt", "apply_kernel_stats.csv", "apply_pmc.txt", "apply_host.txt",
             "strong_cells_per_rank.json", "cfg2_bench.json", "cfg2_kernel_stats.csv", "cfg2_pmc.txt", "photo_phases.txt"):
    src = os.path.join(G, f"{tag}_{name}")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{tag}_{name}"))
rows = list(csv.DictReader(open(os.path.join(P, f"{tag}_pmc_fetch_write.csv"))))
acc = collections.defaultdict(list)
for r in rows:
    acc[(r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("kmg::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
kernels = {}
for (k, c), v in sorted(acc.items()):
    kernels.setdefault(k, {})[c + "_KiB"] = statistics.mean(v)
t = json.load(open(os.path.join(P, "traffic.json")))
stream = 8192 * 8192 * 4
lab = kernels.get("k_labels_pairs") or kernels.get("k_labels")
note = ""
if lab:
    gather = lab["FETCH_SIZE_KiB"] * 1024 - stream / 2
    t["bytes_per_launch"]["k_labels"] = stream + gather + lab["WRITE_SIZE_KiB"] * 1024
    note = (f"k_labels_pairs: FETCH_SIZE {lab['FETCH_SIZE_KiB']*1024/1e6:.1f} MB raw.  The pixel stream (268.4 MB, wide coalesced) is tallied at half "
            f"(134.2 MB, gfx950 correction of MI355X_MICROARCH.md section HBM); the remaining {gather/1e6:.1f} MB are the line fills of the "
            "per-colour label gathers (sub-line accesses, at face value; served by the Infinity Cache, which FETCH_SIZE counts).  "
            f"traffic = 268.4 (pixels) + {gather/1e6:.1f} (table fills) + {lab['WRITE_SIZE_KiB']*1024/1e6:.1f} (labels written, WRITE_SIZE exact) MB "
            "vs 536.9 MB algorithmic.")
cube = {k: v for k, v in kernels.items() if k.startswith("k_cube")}
if cube:
    # the cube kernels read coalesced 16-byte / 4-byte streams: reads = 2 x FETCH_SIZE (same correction), writes at face value
    tot = sum(2 * v["FETCH_SIZE_KiB"] * 1024 + v["WRITE_SIZE_KiB"] * 1024 for v in cube.values())
    t["bytes_per_launch"]["k_cube"] = tot
    note += (f"  Cube pass (k_cube_one + the one-workgroup tail; three launches with hot cells or k > 256): " + ", ".join(f"{k} {2*v['FETCH_SIZE_KiB']*1024/1e6:.1f} MB read (2 x FETCH_SIZE) + {v['WRITE_SIZE_KiB']*1024/1e6:.1f} MB written"
                                                           for k, v in cube.items()) + f" = {tot/1e6:.1f} MB vs 80.5 MB algorithmic (2^24 counts read, 2^24 labels written).")
src = f"profiles/{tag}_pmc_fetch_write.csv via profiles/traffic.json (separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of tools/profile_round.sh, not the run that quotes them)"
t["source"] = {"k_labels": src, "k_cube": src}
t["rounds"][tag] = {"kernels": kernels, "note": note}


def unit_counters(path):
    """{kernel base name: {counter: mean per launch}} of a tools/pmc_*.sh summary"""
    res, cur = {}, None
    if not os.path.exists(path):
        return res
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip().replace("void ", "").replace("kmg::", "").split("<")[0]
            res.setdefault(cur, {})
        elif cur and "mean/launch" in line:
            name, rest = line.split("mean/launch")
            res[cur][name.strip()] = float(rest.split()[0])
    return res


# vector wave-instructions per launch (SQ_INSTS_VALU) of the kernels the bench line quotes: the VALU roof of kernels_roofline
valu = t.setdefault("valu_wave_instructions", {})
units = unit_counters(os.path.join(P, f"{tag}_pmc_units.txt"))
if "k_labels_pairs" in units and "SQ_INSTS_VALU" in units["k_labels_pairs"]:
    valu["k_labels"] = units["k_labels_pairs"]["SQ_INSTS_VALU"]
cube_v = [v["SQ_INSTS_VALU"] for k, v in units.items() if k.startswith("k_cube") and "SQ_INSTS_VALU" in v]
if cube_v:
    valu["k_cube"] = sum(cube_v)
app = unit_counters(os.path.join(P, f"{tag}_apply_pmc.txt"))
if "k_dither_lists" in app and "SQ_INSTS_VALU" in app["k_dither_lists"]:
    valu["find_dither_k64"] = app["k_dither_lists"]["SQ_INSTS_VALU"] + app.get("k_lab_candidates", {}).get("SQ_INSTS_VALU", 0.0)
c2 = unit_counters(os.path.join(P, f"{tag}_cfg2_pmc.txt"))
if c2:
    step = [v for k, v in c2.items() if k.startswith(("k_cube", "k_labels"))]
    if all("SQ_INSTS_VALU" in v for v in step) and step:
        valu["cfg2_step"] = sum(v["SQ_INSTS_VALU"] for v in step)
    if all("FETCH_SIZE" in v and "WRITE_SIZE" in v for v in step) and step:
        # (pixel stream at 2 x FETCH_SIZE as for the headline; the small tables at face value would be lower: an upper figure)
        t["bytes_per_launch"]["cfg2_step"] = sum(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024 for v in step)
# the commit the counters belong to (the newest one that touches the kernels at the time of the profile run): bench.py quotes it as
# roofline.traffic_age, tests/test_profiles_fresh.py fails when the kernels have changed since
import subprocess
t["commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
t["profile_tag"] = tag
t["valu_source"] = f"profiles/{tag}_pmc_units.txt, _apply_pmc.txt, _cfg2_pmc.txt (SQ_INSTS_VALU, mean per launch, separate rocprofv3 --pmc passes)"
json.dump(t, open(os.path.join(P, "traffic.json"), "w"), indent=1)
d = json.load(open(os.path.join(P, f"{tag}_bench.json")))
print(d["value"], d["ms_per_step"], d["roofline"])
print({k: round(v["ms_per_launch"], 4) for k, v in d["kernels"].items()})
print(t["bytes_per_launch"])
print(note)
for r in csv.DictReader(open(os.path.join(P, f"{tag}_kernel_stats.csv"))):
    if "kmg::" in r["Name"]:
        print(f"  rocprof {r['Name'].split('(')[0][:44]:44s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
