"""Count the instructions of a kernel's hot loop in hipcc's device assembly (`hipcc --cuda-device-only -S`).

    python benchmarks/tools/asm_loop_count.py /tmp/mr.s <mangled kernel name substring> [...]

For every matching kernel: the longest backward-branch loop body (label .. s_cbranch back to it), instructions by class
(vector ALU, v_fma_mix, DPP, LDS, buffer/global loads, scalar, waits).  No GPU needed -- this is how a change to the
MSDA kernels' set-up is sized before it is measured."""
import re
import sys
from collections import Counter


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                yield name, body
                name, body = None, []
            else:
                body.append(line.rstrip("\n"))
    if name:
        yield name, body


def classify(op):
    if op.startswith("v_fma_mix"):
        return "v_fma_mix"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def hot_loop(body):
    labels = {}
    insts = []
    for line in body:
        s = line.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        insts.append(s)
    best = (0, 0, 0)
    for i, s in enumerate(insts):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", s) or re.match(r"^s_branch\s+(\.LBB\d+_\d+)", s)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            n = i - labels[m.group(1)] + 1
            if n > best[0]:
                best = (n, labels[m.group(1)], i + 1)
    return insts[best[1]:best[2]]


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    for name, body in kernels(path):
        if pats and not any(p in name for p in pats):
            continue
        loop = hot_loop(body)
        c = Counter(classify(s.split()[0]) for s in loop)
        dpp = sum(1 for s in loop if "quad_perm" in s or "row_" in s)
        trans = sum(1 for s in loop if s.split()[0] in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32"))
        valu = c["valu"] + c["v_fma_mix"]
        print(f"{name}\n  loop {len(loop)} instructions: VALU {valu} (v_fma_mix {c['v_fma_mix']}, DPP {dpp}, transcendental {trans}),"
              f" LDS {c['lds']}, VMEM {c['vmem']}, SALU {c['salu']}, waits {c['wait']}")


if __name__ == "__main__":
    main()
