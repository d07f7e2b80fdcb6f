"""approximate VGPR liveness over a kernel's straight-line ISA (hipcc --save-temps .s): python tools/vgpr_pressure.py file.s kernel-substring
Treats the body as straight-line code (the chain kernels only branch around single stores), first operand(s) = definitions.
Prints the pressure at every 25th MFMA and the peak, so that one can see WHICH phase of a chain kernel holds the registers."""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = [l.strip() for l in lines[start + 1:end] if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
def regs(tok):
    tok = tok.strip()
    m = re.match(r"^[va]\[(\d+):(\d+)\]$", tok)
    if m:
        return set((tok[0], i) for i in range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^[va](\d+)$", tok)
    if m:
        return {(tok[0], int(m.group(1)))}
    return set()
ins = []
for l in body:
    l = l.split(";")[0].strip()
    if not l:
        continue
    op, _, rest = l.partition(" ")
    toks = [t for t in re.split(r",\s*", rest) if t]
    toks = [t.split(" ")[0] for t in toks]
    if op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "s_", "v_cmp", "v_cmpx")) and not op.startswith("v_cmp") :
        d, u = set(), set().union(*[regs(t) for t in toks]) if toks else set()
    elif op.startswith(("v_cmp",)):
        d, u = set(), set().union(*[regs(t) for t in toks]) if toks else set()
    else:
        d = regs(toks[0]) if toks else set()
        u = set().union(*[regs(t) for t in toks[1:]]) if len(toks) > 1 else set()
        if op.startswith("v_mfma") or op.startswith(("v_fmac", "v_mac", "v_lshl_or", "v_cndmask")):   # read-modify-write forms read their destination too
            if op.startswith(("v_fmac", "v_mac")):
                u |= d
    ins.append((op, d, u, l))
live, press = set(), [0] * len(ins)
for i in range(len(ins) - 1, -1, -1):
    op, d, u, _ = ins[i]
    live -= d
    live |= u
    press[i] = len(live)
m = 0
peak = max(range(len(ins)), key=lambda i: press[i])
for i, (op, d, u, l) in enumerate(ins):
    if op.startswith("v_mfma"):
        m += 1
        if m % 25 == 0:
            print(f"mfma {m:5d} line {i:6d} live {press[i]}")
    if op == "s_barrier":
        print(f"   -- s_barrier at mfma {m}, live {press[i]}")
    if i == peak:
        print(f"*** PEAK {press[i]} at instruction {i} (mfma {m}): {l}")
# optional third argument: MFMA index at which the live set is listed by defining instruction
if len(sys.argv) > 3:
    want = int(sys.argv[3])
    m = 0
    at = None
    for i, (op, d, u, l) in enumerate(ins):
        if op.startswith("v_mfma"):
            m += 1
            if m == want:
                at = i
                break
    # recompute live set at `at`
    live = set()
    for i in range(len(ins) - 1, at - 1, -1):
        op, d, u, _ = ins[i]
        live -= d
        live |= u
    lastdef = {}
    for i in range(at):
        for r in ins[i][1]:
            lastdef[r] = i
    groups = {}
    for r in sorted(live):
        i = lastdef.get(r, -1)
        groups.setdefault(i, []).append(r)
    m_at = {}
    m = 0
    for i, (op, *_r) in enumerate(ins):
        if op.startswith("v_mfma"):
            m += 1
        m_at[i] = m
    for i in sorted(groups):
        rs = groups[i]
        print(f"  def@{i:5d} (mfma {m_at.get(i, 0):4d}) x{len(rs):3d}: {ins[i][3][:110] if i >= 0 else 'entry'}")
