"""Instruction histogram of the hottest loop of a kernel in hipcc's device assembly (hipcc -S --cuda-device-only).

usage: python tools/asm_loop_hist.py file.s kernel_substring [--dump]
The loop is the innermost backward branch span that holds the most matrix instructions.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "lds_write"
    if op.startswith("global_load_lds") or op.startswith("buffer_load"):
        return "dma"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_") and "f64" in op:
        return "valu64"
    if op.startswith("v_cvt") or op.startswith("v_pack") or op.startswith("v_perm"):
        return "valu_cvt"
    if op.startswith("v_mov") or op.startswith("v_readfirstlane") or op.startswith("v_readlane"):
        return "valu_mov"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and key in l and l.rstrip().endswith(":") is False and ":" in l and start is None:
            start = i
        elif l.startswith("_Z") and key in l and start is None:
            start = i
    assert start is not None, "kernel not found"
    end = start
    while end < len(lines) and not lines[end].startswith("\t.end_amdhsa_kernel") and not lines[end].startswith(".Lfunc_end"):
        end += 1
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"^\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            lo, hi = labels[m.group(1)], i
            n = sum(1 for x in body[lo:hi] if x.strip().startswith("v_mfma"))
            if best is None or n > best[0] or (n == best[0] and hi - lo < best[2] - best[1]):
                best = (n, lo, hi)
    n, lo, hi = best
    hist = collections.Counter()
    ops = collections.Counter()
    total = 0
    for l in body[lo:hi + 1]:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        op = s.split()[0]
        hist[classify(op)] += 1
        ops[op] += 1
        total += 1
    print("loop lines %d..%d of kernel, %d instructions, %d mfma" % (lo, hi, total, n))
    for k, v in hist.most_common():
        print("  %-10s %4d" % (k, v))
    if "--ops" in sys.argv:
        for k, v in ops.most_common(60):
            print("    %-28s %4d" % (k, v))
    if "--dump" in sys.argv:
        print("\n".join(body[lo:hi + 1]))


if __name__ == "__main__":
    main()
