"""Static count of what a GEMM kernel's epilogue executes, from the assembly hipcc emits (no GPU needed):
     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on --cuda-device-only -S show-o_amd/csrc/gemm3w.hip -o /tmp/gemm3w.s
     python tools/epilogue_isa.py /tmp/gemm3w.s 'gemm3w_kernelILi4ELi6ELi6ELb1ELb0E'
The main loop is the basic block with the most MFMAs that ends in a backward branch; everything after the LAST block that holds an MFMA is
the epilogue.  Per epilogue block: instruction classes (same letters as tools/check_gemm4h_isa.py) and whether the block is the target of a
backward branch (a loop: its count multiplies by the trip count, which this tool cannot know).  Used to size the Q/K/V^T + gelu epilogue
of the [Wqkv ; W1] launch (DESIGN.md, round-4 status: the 35-49 us that launch spends after its last MFMA)."""
import re
import sys
from collections import Counter

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from check_gemm4h_isa import classify  # noqa: E402


def main():
    path, want = sys.argv[1], sys.argv[2]
    name, blocks, cur = None, [], None
    for line in open(path):
        line = line.rstrip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name and want in name and blocks:
                break
            name = m.group(1)
            if want in name:
                cur = ["entry", []]
                blocks = [cur]
            continue
        if name is None or want not in name:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            cur = [m.group(1), []]
            blocks.append(cur)
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.startswith("//"):
            continue
        if t.startswith("s_endpgm"):
            cur[1].append(t)
            continue
        if re.match(r"^[a-z_0-9]+(\s|$)", t):
            cur[1].append(t.split(";")[0].strip())
    if not blocks:
        raise SystemExit(f"no kernel matching {want}")
    print(name)
    order = {b[0]: i for i, b in enumerate(blocks)}
    last_mfma = max(i for i, b in enumerate(blocks) if any(x.startswith("v_mfma") for x in b[1]))
    loop_heads = set()
    for i, b in enumerate(blocks):
        for ins in b[1]:
            if ins.startswith("s_cbranch") or ins.startswith("s_branch"):
                tgt = ins.split()[-1]
                if tgt in order and order[tgt] <= i:
                    loop_heads.add(tgt)
    tot_main = Counter(classify(x) for b in blocks[: last_mfma + 1] for x in b[1])
    print(f"blocks up to the last MFMA: {last_mfma + 1}, instructions {sum(tot_main.values())}: " + " ".join(f"{k}{v}" for k, v in sorted(tot_main.items())))
    tot = Counter()
    print("epilogue blocks (after the last MFMA):")
    for b in blocks[last_mfma + 1:]:
        c = Counter(classify(x) for x in b[1])
        tot.update(c)
        if sum(c.values()) >= 8:
            ops = Counter(x.split()[0] for x in b[1])
            top = ", ".join(f"{k} x{v}" for k, v in ops.most_common(6))
            print(f"  {b[0]:>12} {'LOOP ' if b[0] in loop_heads else '     '}{sum(c.values()):5d}: " + " ".join(f"{k}{v}" for k, v in sorted(c.items())) + f"   [{top}]")
    print(f"epilogue total (each block once): {sum(tot.values())}: " + " ".join(f"{k}{v}" for k, v in sorted(tot.items())))
    ops = Counter(x.split()[0] for b in blocks[last_mfma + 1:] for x in b[1])
    print("most frequent opcodes: " + ", ".join(f"{k} x{v}" for k, v in ops.most_common(25)))


if __name__ == "__main__":
    main()
