"""Print the kernel timeline of one graphed bench step from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o p -- python bench.py --no-cpu-baseline
    python benchmarks/step_timeline.py gpurun_out/tl/p_kernel_trace.csv [step-index]
"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step starts at its first flatten launch (one for the whole pyramid, or one per level back to back)
    first = [i for i, r in enumerate(rows) if "pyramid_flatten" in r["Kernel_Name"]
             and (i == 0 or "pyramid_flatten" not in rows[i - 1]["Kernel_Name"])]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(first) // 2
    s, e = first[k], first[k + 1]
    t0 = int(rows[s]["Start_Timestamp"])
    prev, tot = t0, 0
    for r in rows[s:e]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].replace("void ", "").replace("at::native::", "")
        if n.startswith("Cijk"):
            n = "GEMM " + n[5:28] + " " + n.split("UserArgs_")[1][:14]
        print("%8.1f gap %5.1f dur %6.1f  %s  grid=%s" % ((st - t0) / 1e3, (st - prev) / 1e3, (en - st) / 1e3, n[:90],
                                                          r.get("Grid_Size_X", "")))
        prev = en
        tot += en - st
    print("kernel sum us", tot / 1e3, "span", (prev - t0) / 1e3, "n", e - s)


if __name__ == "__main__":
    main()
