#!/usr/bin/env python
"""tools/step_timeline.py -- where a CD-1 step's time goes BETWEEN its launches: reads the kernel trace of

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline

and prints, over all complete steps (k_hidden, k_visible, k_hidden, k_grad in a row), the median duration of each launch and
the median gap between the end of one launch and the start of the next (same clock: the dispatch timestamps of one queue).

    python tools/step_timeline.py DIR
"""
import csv
import glob
import json
import statistics
import sys


def main(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    def kind(n):
        for k in ("k_hidden", "k_visible", "k_grad"):
            if k in n:
                return k
        return "other"
    seq = [(s, e, kind(n)) for s, e, n in rows]
    want = ["k_hidden", "k_visible", "k_hidden", "k_grad"]
    dur = [[] for _ in want]
    gap = [[] for _ in want]      # gap[i]: end of launch i -> start of launch i + 1 (gap[3]: to the next step's first launch)
    step = []
    i = 0
    while i + 4 < len(seq):
        if [k for _, _, k in seq[i:i + 4]] == want and seq[i + 4][2] == "k_hidden":
            for j in range(4):
                dur[j].append(seq[i + j][1] - seq[i + j][0])
                gap[j].append(seq[i + j + 1][0] - seq[i + j][1])
            step.append(seq[i + 4][0] - seq[i][0])
            i += 4
        else:
            i += 1
    med = lambda x: round(statistics.median(x) / 1e3, 3) if x else None
    print(json.dumps({"what": "CD-1 step timeline from the kernel trace (us, medians over %d steps)" % len(step),
                      "launch_us": dict(zip(["k_hidden#1", "k_visible", "k_hidden#2", "k_grad"], map(med, dur))),
                      "gap_after_us": dict(zip(["k_hidden#1", "k_visible", "k_hidden#2", "k_grad"], map(med, gap))),
                      "step_start_to_start_us": med(step)}))


if __name__ == "__main__":
    main(sys.argv[1])
