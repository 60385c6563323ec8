#!/usr/bin/env python
"""HBM traffic of every launch of one C2 step, in launch order: rocprofv3 --pmc FETCH_SIZE (x2 on gfx950, MI355X guide) and
--pmc WRITE_SIZE in separate passes of `bench.py --probe c2 --probe-steps 2`, the LAST step's dispatches, against the launch's
algorithmic bytes (profile records of tools/profile_step.py, same order).   python tools/step_traffic.py [extra bench args] > out.txt"""
import csv, glob, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONE_TIME = ("bias32_build", "pack_kernel", "rocclr", "at::", "fill", "elementwise", "Cijk", "vectorized")
def run(counter):
    tmp = tempfile.mkdtemp(prefix="kvq_st_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", "c", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--probe", "c2", "--probe-steps", "2"] + sys.argv[1:]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    f = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        sys.exit(f"rocprofv3 --pmc {counter} failed: {(r.stderr or r.stdout)[-400:]}")
    rows = [(int(x["Dispatch_Id"]), x["Kernel_Name"], float(x["Counter_Value"])) for x in csv.DictReader(open(f[0])) if x["Counter_Name"] == counter]
    shutil.rmtree(tmp, ignore_errors=True)
    agg = {}
    for d, n, v in rows:                      # one row per XCD / instance: sum per dispatch
        agg.setdefault(d, [n, 0.0]); agg[d][1] += v
    out = [(d, n, v) for d, (n, v) in sorted(agg.items()) if not any(k in n for k in ONE_TIME)]
    return out
f, w = run("FETCH_SIZE"), run("WRITE_SIZE")
assert len(f) == len(w), (len(f), len(w))
n = len(f) // 2                               # two steps: keep the second
f, w = f[-n:], w[-n:]
def short(nm):
    nm = re.sub(r"^void ", "", nm).replace("kvq::", ""); return re.sub(r"\(.*$", "", nm)[:58]
print(f"{'#':>3} {'kernel':58} {'fetch MB':>9} {'write MB':>9} {'total MB':>9}")
tf = tw = 0.0
fam = {}
for i, ((_, nm, fv), (_, _, wv)) in enumerate(zip(f, w)):
    fb, wb = 2.0 * 1024.0 * fv / 1e6, 1024.0 * wv / 1e6
    tf += fb; tw += wb
    k = re.sub(r"<.*", "", short(nm)); fam[k] = fam.get(k, 0.0) + fb + wb
    print(f"{i:3d} {short(nm):58} {fb:9.1f} {wb:9.1f} {fb + wb:9.1f}")
print(f"step: fetch {tf:.1f} MB + write {tw:.1f} MB = {tf + tw:.1f} MB over {n} launches")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]): print(f"   {k:40} {v:9.1f} MB")
