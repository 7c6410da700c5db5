"""Instruction mix of the MFMA-carrying basic blocks of one kernel in a hipcc -S listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -I../../include -o /tmp/attn.s attn.hip
    python3 tools/isa_mix.py /tmp/attn.s attn_bf16ILi64E
"""
import collections
import re
import sys


def main(path, needle, min_mfma=8):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if needle in l and re.match(r"^_Z\S+:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur, name = [], [], "entry"
    for l in lines[start + 1:end]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append((name, cur))
            name, cur = l.split(":")[0], []
        else:
            cur.append(l.strip())
    blocks.append((name, cur))
    for n, b in blocks:
        ins = [x.split()[0] for x in b if x and not x.startswith((";", "."))]
        mf = sum(x.startswith("v_mfma") for x in ins)
        if mf >= min_mfma:
            c = collections.Counter(ins)
            valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
            print(f"{n}: {len(ins)} instructions, {mf} MFMA, {valu} other vector, {sum(v for k, v in c.items() if k.startswith('ds_'))} LDS")
            for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
                print(f"    {k:28s} {v}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8)
