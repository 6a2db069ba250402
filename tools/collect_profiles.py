"""Copies the judged summaries of a tools/prof_round.sh run from gpurun_out/<tag>_* into profiles/<round>_*
and rewrites profiles/hbm_traffic.json from their PMC passes (2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes).
usage: python tools/collect_profiles.py <tag> [round-prefix, default r03]"""
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def first(pat):
    f = sorted(glob.glob(pat, recursive=True))
    return f[0] if f else None


def box_of(wl):
    """The lease's identity record (tools/prof.sh writes it beside every trace): host, GPU uuid, library and bench hashes."""
    f = os.path.join(go, "%s_%s" % (tag, wl), "box.json")
    return open(f).read().strip() if os.path.isfile(f) else "{}"


def own_line(wl):
    """ms_per_step / kernel_ms / frac of the line the TRACED run itself printed (same box, same library, under the tracer)."""
    f = os.path.join(go, "%s_%s" % (tag, wl), "trace.log")
    if not os.path.isfile(f):
        return ""
    for line in open(f):
        if line.startswith("{"):
            try:
                d = json.loads(line)
                rf = d.get("roofline", {})
                return "ms_per_step=%s kernel_ms=%s frac=%s value=%s" % (d.get("ms_per_step"), rf.get("kernel_ms"), rf.get("frac"), d.get("value"))
            except ValueError:
                pass
    return ""


for wl in ("acq", "acq59", "acq10ms", "wf14", "ddc14", "cfg2_chain", "receivers", "receivers_light"):
    ks = first(os.path.join(go, "%s_%s" % (tag, wl), "trace", "**", "*kernel_stats.csv"))
    if ks:
        dst = os.path.join(pr, "%s_%s_kernel_stats.csv" % (rnd, wl))
        shutil.copy(ks, dst)
        # the identity of the lease as a last row (name column; numeric columns zero so that CSV readers keep working)
        ncol = len(open(dst).readline().split(",")) - 1
        with open(dst, "a") as fh:
            fh.write('"# box %s | traced run: %s"%s\n' % (box_of(wl).replace('"', "'"), own_line(wl), ",0" * ncol))
    sm = os.path.join(go, "%s_%s.summary.txt" % (tag, wl))
    if os.path.isfile(sm):
        shutil.copy(sm, os.path.join(pr, "%s_%s_summary.txt" % (rnd, wl)))


def counters(wl, kernel_sub):
    """mean per dispatch of the named counters for the kernel whose name contains kernel_sub"""
    text = open(os.path.join(pr, "%s_%s_summary.txt" % (rnd, wl))).read()
    out = {}
    for block in re.split(r"\n  (?=\S)", text):
        if kernel_sub not in block.split("\n")[0]:
            continue
        for m in re.finditer(r"^\s+(\w+)\s+([0-9.]+)\s+\(n=", block, re.M):
            out.setdefault(m.group(1), float(m.group(2)))
    return out


def entry(wl, sub, label):
    c = counters(wl, sub)
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        raise SystemExit("no FETCH_SIZE / WRITE_SIZE for %s in %s" % (sub, wl))
    return {"kernel": label, "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
            "bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
            "tcc_hit": c.get("TCC_HIT_sum"), "tcc_miss": c.get("TCC_MISS_sum"),
            "source": "profiles/%s_%s_summary.txt" % (rnd, wl)}


tab = {"_how": "tools/prof_round.sh on MI355X: separate rocprofv3 --pmc passes of bench.py (pass 3: FETCH_SIZE; pass 4: "
               "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum), mean per dispatch; bytes = 2 x FETCH_SIZE KB (gfx950 tallies "
               "128-byte read requests at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KB.  bench.py measures "
               "the same live (live_traffic_passes); this table is its fallback",
       "_source": "profiles/%s_*_summary.txt" % rnd}
tab["acq"] = {"32": entry("acq", "acq_correlate_kernel<4, 1,", "acq_correlate_kernel<4, 1, true, false>")}
tab["wf14"] = {"28672": entry("wf14", "wf_frame_kernel", "wf_frame_kernel<false>")}
p1 = entry("acq10ms", "acq_correlate_kernel<16, 1,", "acq_correlate_kernel<16, 1, true, false>")
p4 = entry("acq10ms", "acq_correlate8_kernel<16", "acq_correlate8_kernel<16, false>")
tab["acq10ms"] = {"2": {"kernel": "acq_correlate_kernel<16,1> + acq_correlate8_kernel<16>",
                        "bytes_per_launch": p1["bytes_per_launch"] + p4["bytes_per_launch"], "parts": [p1, p4]}}
q1 = entry("acq59", "acq_correlate_kernel<4, 1,", "acq_correlate_kernel<4, 1, true, false>")
q4 = entry("acq59", "acq_correlate8_kernel<4", "acq_correlate8_kernel<4, false>")
tab["acq59"] = {"32": {"kernel": "acq_correlate_kernel<4,1> + acq_correlate8_kernel<4>",
                       "bytes_per_launch": q1["bytes_per_launch"] + q4["bytes_per_launch"], "parts": [q1, q4]}}
json.dump(tab, open(os.path.join(pr, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: (v if k.startswith("_") else {kk: vv.get("bytes_per_launch") for kk, vv in v.items()}) for k, v in tab.items()}, indent=1))
