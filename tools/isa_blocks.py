"""Basic blocks of one kernel in a hipcc -S listing, in order: instruction, MFMA, other-vector, LDS, exp, v_mov and scratch counts.
    python3 tools/isa_blocks.py /tmp/attn16.s attn_bf16_m16pILb0 [min_instructions]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
needle = sys.argv[2]
min_ins = int(sys.argv[3]) if len(sys.argv) > 3 else 30
start = next(i for i, l in enumerate(lines) if needle in l and re.match(r"^_Z\S+:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
name, cur, blocks = "entry", [], []
for l in lines[start + 1:end]:
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append((name, cur))
        name, cur = l.split(":")[0], []
    else:
        cur.append(l.strip())
blocks.append((name, cur))
for n, b in blocks:
    ins = [x.split()[0] for x in b if x and not x.startswith((";", "."))]
    c = collections.Counter(ins)
    mf = sum(v for k, v in c.items() if k.startswith("v_mfma"))
    if mf == 0 and len(ins) < min_ins:
        continue
    valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
    print(f"{n:10s} {len(ins):4d} ins  mfma {mf:3d}  valu {valu:3d}  lds {sum(v for k, v in c.items() if k.startswith('ds_')):3d}  exp {c.get('v_exp_f32_e32', 0):3d}"
          f"  mov {c.get('v_mov_b32_e32', 0):3d}  scratch {sum(v for k, v in c.items() if k.startswith('scratch')):3d}  nop {c.get('s_nop', 0)}")
