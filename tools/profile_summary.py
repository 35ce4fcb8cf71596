#!/usr/bin/env python3
"""tools/profile_summary.py RAW_DIR TAG CONFIG -- condense the rocprofv3 output of tools/profile_round.sh into the files committed under
profiles/: RND_<tag>_kernel_stats.csv (rocprofv3's own --stats table), RND_<tag>_bench.json (the bench line of the traced run),
RND_traffic_<config>.json (HBM bytes per pass from FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes) and
RND_<tag>_insts.json (VALU / SALU instructions per kernel and per secondary ray).  RND = $RTW_ROUND (default r03).  Every file records the hash of the
kernels it was taken on (`kernels_sha`, from the traced run's rtw_version()): bench.py prints a figure from here only when that hash is the loaded library's."""
import csv
import glob
import json
import os
import re
import shutil
import sys

raw, tag, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
RND = os.environ.get("RTW_ROUND", "r03")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
os.makedirs(PROF, exist_ok=True)


def short(name):
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:40]


def find(sub, pattern):
    f = glob.glob(os.path.join(raw, sub, "**", pattern), recursive=True)
    return f[0] if f else None


def bench_line(sub):
    p = os.path.join(raw, sub + ".bench.json")
    try:
        return json.loads(open(p).read())
    except Exception:
        return None


def timed_dispatch_window(trace_csv, steps):
    """the dispatches of the TIMED rtw_render_passes call: bench.py renders 1 + 1 + warmup passes, then the K timed ones, then untimed replays; the timed call is the
    first run of pass-batched kernels that follows the warm-up calls -- identified as the groups between the 3rd gresolve... simpler: all dispatches are summed per
    kernel over the window [first gprimary after the warm-up's last gresolve, the gresolve that closes the timed groups]"""
    rows = list(csv.DictReader(open(trace_csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    return rows, names


def window_of_timed_call(names, k_steps, warm):
    """indices [a, b) of the timed call's dispatches: bench.py zeroes the two framebuffer tensors (two torch fill kernels) right before the timed
    rtw_render_passes call and replays the passes through render_kernel (pipeline 0) right after it; everything between is the timed call."""
    replay = [i for i, n in enumerate(names) if n == "render_kernel"]
    b = replay[0] if replay else len(names)
    fills = [i for i, n in enumerate(names[:b]) if "elementwise" in n]
    a = fills[-1] + 1 if fills else 0
    while b > a and not (names[b - 1].startswith("g") or "primary" in names[b - 1] or "shade" in names[b - 1] or "trace" in names[b - 1] or "resolve" in names[b - 1]):
        b -= 1
    # bench.py repeats the timed call (`repeat_calls` times, untimed, each after a synchronise) before the replay: the window is cut after the
    # timed call's own last gresolve -- the calls launch the same kernels, so the timed one holds 1 / (1 + repeats) of the window's gresolve launches
    res = [i for i in range(a, b) if names[i] == "gresolve_kernel"]
    if REPEATS > 0 and res and len(res) % (1 + REPEATS) == 0:
        b = res[len(res) // (1 + REPEATS) - 1] + 1
    return a, b


out = {}
bl = bench_line("kt")
steps = bl["steps"] if bl else 20
REPEATS = int((bl or {}).get("repeat_calls", 3 if (bl or {}).get("repeat_call_ms_per_step") else 0))
warm = bl["warmup"] if bl else 5


def sha_of(line):
    v = (line or {}).get("library", "")
    return v.split("kernels ")[1].split(")")[0].strip() if "kernels " in v else None


KSHA = sha_of(bl)
out["kernels_sha"] = KSHA
if bl:
    json.dump(bl, open(os.path.join(PROF, "%s_%s_bench.json" % (RND, tag)), "w"), indent=1)
st = find("kt", "*kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(PROF, "%s_%s_kernel_stats.csv" % (RND, tag)))
# per-kernel durations inside the timed call
kt = find("kt", "*kernel_trace.csv")
if kt:
    rows, names = timed_dispatch_window(kt, steps)
    a, b = window_of_timed_call(names, steps, warm)
    per = {}
    for r, n in list(zip(rows, names))[a:b]:
        d = per.setdefault(n, [0, 0.0])
        d[0] += 1
        d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    t0, t1 = int(rows[a]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows[a:b])
    out["timed_call"] = {"steps": steps, "span_us": (t1 - t0) / 1e3, "span_us_per_step": (t1 - t0) / 1e3 / steps,
                         "kernels": {n: {"launches": c, "total_us": t, "us_per_step": t / steps} for n, (c, t) in sorted(per.items(), key=lambda x: -x[1][1])}}


def counters(sub):
    f = find(sub, "*counter_collection.csv")
    tr = find(sub, "*kernel_trace.csv")
    if not f or not tr:
        return None
    rows, names = timed_dispatch_window(tr, steps)
    a, b = window_of_timed_call(names, steps, warm)
    ids = set(r["Dispatch_Id"] for r in rows[a:b])
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Dispatch_Id"] not in ids:
            continue
        d = per.setdefault(short(r["Kernel_Name"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return per


fe, wr, ins = counters("fetch"), counters("write"), counters("insts")
if fe and wr:
    STREAMING = ("gsky_kernel", "gresolve_kernel", "primary_sky_kernel")      # 16 B per lane, consecutive lanes: gfx950 tallies their 128-B requests as 64 B
    per_kernel, total = {}, 0.0
    for k in sorted(set(fe) | set(wr)):
        rd = fe.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0
        if k in STREAMING:
            rd *= 2.0
        wb = wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
        per_kernel[k] = {"hbm_read_bytes_per_pass": rd / steps, "hbm_write_bytes_per_pass": wb / steps, "fetch_doubled": k in STREAMING}
        total += (rd + wb) / steps
    traffic = {"kernels_sha": sha_of(bench_line("fetch")) or KSHA, "workload": bl["config"]["workload"] if bl else cfg, "steps_in_the_timed_call": steps, "hbm_bytes_per_pass": total, "per_kernel": per_kernel,
               "ms_per_step_of_the_traced_run": bl["ms_per_step"] if bl else None,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `bench.py --config %s --no-cpu --steps %d --warmup %d` (tools/profile_round.sh); "
                         "KB * 1024 summed over the dispatches of the timed rtw_render_passes call, divided by its passes; FETCH doubled for the kernels that stream 16 B per lane "
                         "(gfx950 tallies 128-B requests as 64 B, MI355X_MICROARCH.md 'HBM'); gather-style kernels left uncorrected (uncalibrated width); Infinity-Cache hits are "
                         "counted by these counters, so this is an upper bound of what reaches HBM" % (cfg, steps, warm)}
    json.dump(traffic, open(os.path.join(PROF, "%s_traffic_%s.json" % (RND, cfg)), "w"), indent=1)
    out["hbm_bytes_per_pass"] = total
if ins:
    il = bench_line("insts") or bl
    sec = None
    if il:
        c = il["roofline"]["counters_per_pass_as_run"]
        sec = (c["rays"] - c["camera_rays"]) * steps
    tr = {k: v for k, v in ins.items() if k.startswith("gtrace") or k.startswith("trace_wave")}
    valu = sum(v.get("SQ_INSTS_VALU", 0.0) for v in tr.values())
    salu = sum(v.get("SQ_INSTS_SALU", 0.0) for v in tr.values())
    res = {"kernels_sha": sha_of(il) or KSHA, "per_kernel_in_the_timed_call": ins, "steps_in_the_timed_call": steps, "secondary_rays_in_the_timed_call": sec,
           "valu_instructions_per_pass": sum(v.get("SQ_INSTS_VALU", 0.0) for v in ins.values()) / steps,
           "trace_kernels_valu_instructions_per_secondary_ray": valu / sec if sec else None,
           "trace_kernels_salu_instructions_per_secondary_ray": salu / sec if sec else None,
           "note": "SQ_INSTS_VALU / SQ_INSTS_SALU are per-wave instruction counts summed over the dispatch; divided by the secondary rays the timed call traced (bench counters)"}
    json.dump(res, open(os.path.join(PROF, "%s_%s_insts.json" % (RND, tag)), "w"), indent=1)
    out["valu_per_secondary_ray"] = res["trace_kernels_valu_instructions_per_secondary_ray"]
json.dump(out, open(os.path.join(PROF, "%s_%s_timeline.json" % (RND, tag)), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "timed_call"}))
if "timed_call" in out:
    print("span per step %.1f us" % out["timed_call"]["span_us_per_step"])
    for n, d in out["timed_call"]["kernels"].items():
        print("  %-28s launches %3d  %.1f us/step" % (n, d["launches"], d["us_per_step"]))
