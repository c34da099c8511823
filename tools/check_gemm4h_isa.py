"""Prints the instruction stream of a kernel's hot loops as one letter per instruction (from the .s file hipcc -save-temps leaves):
M mfma, R ds_read, D LDS-DMA (buffer/global_load ... lds), w s_waitcnt, B s_barrier, s SALU, v VALU, n s_nop, X scratch / spill traffic,
L other loads, T stores, a v_accvgpr moves.  Used to check that gemm4h's k-loop is what the source's sched_group_barrier pipelines ask for
(reads and DMAs between MFMAs, no spill traffic, only the hand-placed waits).   python tools/check_gemm4h_isa.py file.s [kernel-substr]"""
import re
import sys


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"):
        return "M"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "R"
    if ("buffer_load" in op or "global_load" in op) and " lds" in ins:
        return "D"
    if op.startswith("scratch_") or (op.startswith("buffer_") and "offen" not in ins and "s[0:3]" in ins and " lds" not in ins):
        return "X"
    if op.startswith("s_waitcnt"):
        return "w"
    if op.startswith("s_barrier"):
        return "B"
    if op.startswith("s_nop"):
        return "n"
    if op.startswith("v_accvgpr"):
        return "a"
    if op.startswith("ds_write") or op.startswith("ds_store") or op.startswith("global_store") or op.startswith("buffer_store"):
        return "T"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "L"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "J"
    if op.startswith("s_"):
        return "s"
    if op.startswith("v_"):
        return "v"
    return "?"


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    verbose = len(sys.argv) > 3
    cur, blocks, name = None, [], None
    for line in open(path):
        line = line.rstrip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            cur = None
        if name is None or want not in name:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            cur = [m.group(1), [], name]
            blocks.append(cur)
            continue
        if cur is None:
            if re.match(r"^_Z\w+:", line):
                cur = ["entry", [], name]
                blocks.append(cur)
            continue
        s = line.strip()
        if not s or s.startswith(";") or s.startswith(".") or s.startswith("//"):
            continue
        s = s.split(";")[0].strip()
        if s:
            cur[1].append(s)
    for label, ins, kn in blocks:
        cl = "".join(classify(i) for i in ins)
        nm = cl.count("M")
        if nm < 32:
            continue
        print(f"== {kn[:60]} {label}: {len(ins)} instructions, {nm} MFMA, {cl.count('R')} ds_read, {cl.count('D')} DMA, {cl.count('X')} scratch, "
              f"{cl.count('w')} waitcnt, {cl.count('B')} barrier, {cl.count('v')} VALU, {cl.count('s')} SALU, {cl.count('n')} nop, {cl.count('a')} accvgpr")
        for i in range(0, len(cl), 128):
            print("   ", cl[i:i + 128])
        if verbose:
            for i in ins:
                if classify(i) in "wXn":
                    print("      ", i)


if __name__ == "__main__":
    main()
