"""GPU timeline of a training run from a rocprofv3 kernel trace: per step (delimited by the fused Adam kernel) the span, the
busy time per queue and the idle gaps on the main queue with the kernel that ended each gap.
    python tools/trace_gaps.py <kernel_trace.csv> [first_step] [n_steps]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    m = re.search(r"(k_\w+|multi_tensor_apply|vectorized_gather|index_elementwise|reduce_kernel|elementwise|fillBuffer|copyBuffer)", n)
    r["n"] = m.group(1) if m else n[:40]
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if "FusedAdam" in r["Kernel_Name"] or "k_adam" in r["Kernel_Name"]]   # torch's fused Adam | the library's
print("kernels", len(rows), "adam launches", len(adam), "queues", collections.Counter(r["Queue_Id"] for r in rows))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 20
mainq = rows[adam[0]]["Queue_Id"]
tot = collections.Counter(); gapby = collections.Counter(); gapn = collections.Counter()
span = busy_main = busy_other = 0
for k in range(first, first + nst):
    a, b = adam[k], adam[k + 1]
    seg = rows[a + 1:b + 1]
    t0, t1 = rows[a]["e"], rows[b]["e"]
    span += t1 - t0
    prev_end = t0
    for r in seg:
        if r["Queue_Id"] == mainq:
            busy_main += r["e"] - r["s"]
            g = r["s"] - prev_end
            if g > 0:
                gapby[r["n"]] += g; gapn[r["n"]] += 1
            prev_end = max(prev_end, r["e"])
        else:
            busy_other += r["e"] - r["s"]
        tot[r["n"]] += r["e"] - r["s"]
print("per step: span %.3f ms  main-queue busy %.3f ms  other-queue busy %.3f ms  main-queue idle %.3f ms  kernels/step %.0f" % (
    span / nst / 1e6, busy_main / nst / 1e6, busy_other / nst / 1e6, (span - busy_main) / nst / 1e6, sum(1 for _ in rows[adam[first]:adam[first + nst]]) / nst))
print("idle gaps on the main queue, by the kernel that follows (us/step, count/step, avg us):")
for n, g in gapby.most_common(25):
    print("   %-28s %8.1f %6.1f %7.2f" % (n, g / nst / 1e3, gapn[n] / nst, g / gapn[n] / 1e3))
print("kernel time (us/step):")
for n, g in tot.most_common(30):
    print("   %-28s %8.1f" % (n, g / nst / 1e3))
if "--seq" in sys.argv:   # the main queue's launches of one step in order: start offset, duration, gap before, grid, name
    k = first + nst // 2
    a, b = adam[k], adam[k + 1]
    t0 = rows[a]["e"]
    prev = t0
    print("sequence of step %d on the main queue (us from the previous Adam's end):" % k)
    i = 0
    for r in rows[a + 1:b + 1]:
        if r["Queue_Id"] != mainq:
            continue
        i += 1
        print("  %3d  t=%7.1f  dur=%6.1f  gap=%5.1f  grid=%-8s %s" % (i, (r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, (r["s"] - prev) / 1e3,
                                                                    r.get("Grid_Size", "?"), r["Kernel_Name"][:90]))
        prev = r["e"]
