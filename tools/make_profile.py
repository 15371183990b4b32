"""Turn rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/.

usage: python tools/make_profile.py <tag> <stats_dir> [<fetch_dir> <write_dir>] [--cmd "..."] [--bench-log file]
"""
import argparse, collections, csv, glob, json, os

ap = argparse.ArgumentParser()
ap.add_argument("tag"); ap.add_argument("stats_dir"); ap.add_argument("fetch_dir", nargs="?"); ap.add_argument("write_dir", nargs="?")
ap.add_argument("--cmd", default=""); ap.add_argument("--bench-log", default=""); ap.add_argument("--note", default="")
a = ap.parse_args()

def one(pat):
    g = glob.glob(pat)
    return g[0] if g else None

out_md = "profiles/%s_kernel_stats.md" % a.tag
rows = list(csv.DictReader(open(one(a.stats_dir + "/*/*kernel_stats.csv"))))
bench = ""
if a.bench_log:
    lines = [l for l in open(a.bench_log) if l.startswith("{")]
    bench = lines[-1].strip() if lines else ""
with open(out_md, "w") as f:
    f.write("# %s: rocprofv3 --kernel-trace --stats\n\n" % a.tag)
    if a.cmd: f.write("Command (MI355X box): `%s`\n\n" % a.cmd)
    if a.note: f.write(a.note + "\n\n")
    if bench: f.write("bench line of the profiled run:\n\n```\n%s\n```\n\n" % bench)
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows:
        f.write("| `%s` | %s | %.3f | %.2f | %s |\n" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
print("wrote", out_md)

if a.fetch_dir and a.write_dir:
    def agg(d, cname):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(one(d + "/*/*counter_collection.csv"))):
            if r["Counter_Name"] == cname:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
        return acc
    fe, wr = agg(a.fetch_dir, "FETCH_SIZE"), agg(a.write_dir, "WRITE_SIZE")
    pmc = {}
    with open("profiles/%s_pmc.md" % a.tag, "w") as f:
        f.write("# %s: HBM traffic per launch from PMC counters (separate --pmc passes, MI355X_MICROARCH.md HBM section)\n\n" % a.tag)
        f.write("FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read\n"
                "(16 B per lane), so the corrected read volume is 2 x FETCH_SIZE for the kernels below whose loads are 16 B per lane\n"
                "(recurrent kernels: gate activations; GEMMs: operand tiles); WRITE_SIZE is taken as is.\n\n")
        f.write("| kernel | launches | FETCH_SIZE avg [KB] | WRITE_SIZE avg [KB] | corrected bytes per launch (2*F + W) |\n|---|---|---|---|---|\n")
        for k in sorted(fe, key=lambda k: -sum(fe[k])):
            fa = sum(fe[k]) / len(fe[k]); wa = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0])))
            tot = (2 * fa + wa) * 1024
            pmc[k] = {"fetch_kb": fa, "write_kb": wa, "bytes_per_launch": tot, "launches": len(fe[k])}
            f.write("| `%s` | %d | %.1f | %.1f | %.3e |\n" % (k[:70], len(fe[k]), fa, wa, tot))
    json.dump(pmc, open("profiles/%s_pmc.json" % a.tag, "w"), indent=1)
    print("wrote pmc")
