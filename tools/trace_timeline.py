#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: wall time, GPU-busy union, per-queue busy time, idle gaps
and the kernels ranked by time (their duration as it was in the overlapped step).
    python tools/trace_timeline.py <dir or kernel_trace.csv> [n_steps]"""
import collections, csv, glob, os, re, sys

def canon(name):
    name = re.sub(r"^void\s+", "", name).replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*\)$", "", name)
    m = re.search(r"gemm_kernelILb(\d)ELb(\d)E(DF16b|f)Lb(\d)ELi(n?\d+)E", name)
    if m:
        b = lambda x: "t" if x == "1" else "f"
        return f"gemm<{b(m.group(1))},{b(m.group(2))},{'bf16' if m.group(3) == 'DF16b' else 'f32'},{b(m.group(4))},{m.group(5)}>"
    m = re.search(r"panel_kernelILi(\d+)ELb(\d)ELi(\d)E", name)
    if m:
        return f"panel<{m.group(1)},{'bwd' if m.group(3) == '1' else 'fwd'}>"
    return name[:60]

path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), canon(r["Kernel_Name"])))
rows.sort()
# steps are delimited by the optimizer kernel (adamw_ema): take the LAST complete step
ends = [i for i, r in enumerate(rows) if r[3].startswith("center_update")]
if len(ends) < 2:
    sys.exit("need at least two steps in the trace")
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t0, t1 = step[0][0], max(r[1] for r in step)
print(f"step: {len(step)} kernels, wall {1e-6 * (t1 - t0):.3f} ms, sum of kernel durations {1e-6 * sum(r[1] - r[0] for r in step):.3f} ms")
ev = sorted([(r[0], 1) for r in step] + [(r[1], -1) for r in step])
busy = 0; depth = 0; last = t0; conc = collections.Counter()
for t, d in ev:
    if depth > 0: busy += t - last
    conc[min(depth, 3)] += t - last
    depth += d; last = t
print(f"GPU busy (union) {1e-6 * busy:.3f} ms, idle {1e-6 * (t1 - t0 - busy):.3f} ms; time with 1 / 2 / 3+ kernels in flight: " +
      " / ".join(f"{1e-6 * conc[k]:.2f}" for k in (1, 2, 3)) + " ms")
perq = collections.defaultdict(int)
for r in step: perq[r[2]] += r[1] - r[0]
print("per queue busy (ms):", {q: round(1e-6 * v, 2) for q, v in sorted(perq.items(), key=lambda kv: -kv[1])})
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    agg[r[3]][0] += 1; agg[r[3]][1] += r[1] - r[0]
print(f"{'kernel':58s} {'n':>4s} {'ms':>7s} {'avg us':>8s}")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{k:58s} {n:4d} {1e-6 * t:7.3f} {1e-3 * t / n:8.1f}")
# phase marks
def first(pred): return next((r[0] for r in step if pred(r[3])), None)
marks = {"loss": first(lambda k: k.startswith("row_stats")), "optimizer": first(lambda k: k.startswith("sumsq") or k.startswith("adamw"))}
for k, v in marks.items():
    if v: print(f"{k} starts at +{1e-6 * (v - t0):.3f} ms")
